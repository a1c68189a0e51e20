"""Run-directory helpers with the call shape of the reference's utils/directory.py:12-42 (mkdir, mkdir_join):
`mkdir_join(root, 'ctc', label_type, model.name)` creates each level that is not a file name (no '.') and returns
the joined path; a None root is passed through."""
import os


def mkdir(path_to_dir):
    if path_to_dir is not None:
        os.makedirs(path_to_dir, exist_ok=True)
    return path_to_dir


def mkdir_join(path_to_dir, *dir_name):
    if path_to_dir is None:
        return None
    path = path_to_dir
    for part in dir_name:
        path = os.path.join(path, part)
        if '.' not in part:            # a component with a dot is taken for a file name and not created
            mkdir(path)
    return path
