"""Phone <-> index maps -- mirror of utils/io/labels/phone.py:11-70 (Phone2idx, Idx2phone).
Map file: one `<phone>  <index>` pair per line (examples/timit/metrics/mapping_files/phone61.txt)."""
import numpy as np


def _read_map(map_file_path):
    pairs = []
    with open(map_file_path, 'r') as f:
        for line in f:
            tok = line.strip().split()
            if len(tok) >= 2:
                pairs.append((str(tok[0]), int(tok[1])))
    return pairs


class Phone2idx(object):
    def __init__(self, map_file_path):
        self.map_dict = dict(_read_map(map_file_path))

    def __call__(self, phone_list):
        """list of phone strings -> np.ndarray of indices (KeyError on an unknown phone, as the reference)."""
        return np.array([self.map_dict[p] for p in phone_list])


class Idx2phone(object):
    def __init__(self, map_file_path):
        self.map_dict = dict((i, p) for p, i in _read_map(map_file_path))

    def __call__(self, index_list, padded_value=-1):
        """np.ndarray of indices -> 'p1 p2 ...'.  As in the reference (:61-62) the entries equal to -1 are
        the ones dropped, whatever `padded_value` says."""
        assert type(index_list) == np.ndarray, 'index_list should be np.ndarray.'
        return ' '.join(self.map_dict[int(i)] for i in index_list if i != -1)
