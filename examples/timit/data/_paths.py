"""Corpus root and utterance lists shared by the TIMIT dataset classes
(examples/timit/data/load_dataset_*.py:64-96 of the reference)."""
import os
import pickle
from os.path import isfile, join

import numpy as np

DEFAULT_ROOTS = ['/data/inaguma/timit', '/n/sd8/inaguma/corpus/timit/dataset']


def utterance_paths(data_type, label_types, sort_utt, dataset_root=None):
    """-> (input_paths, [label_paths per label type], frame_num_dict), ordered by name or by frame count."""
    roots = [r for r in [dataset_root, os.environ.get('TIMIT_DATASET_ROOT')] if r] + DEFAULT_ROOTS
    for root in roots:
        input_path = join(root, 'inputs', data_type)
        if isfile(join(input_path, 'frame_num.pickle')):
            break
    else:
        raise IOError('frame_num.pickle not found under any of %s (inputs/%s/)' % (roots, data_type))
    with open(join(input_path, 'frame_num.pickle'), 'rb') as f:
        frame_num_dict = pickle.load(f)
    axis = 1 if sort_utt else 0
    names = [n for n, _ in sorted(frame_num_dict.items(), key=lambda x: x[axis])]
    input_paths = np.array([join(input_path, n + '.npy') for n in names])
    label_paths = [np.array([join(root, 'labels', data_type, lt, n + '.npy') for n in names]) for lt in label_types]
    return input_paths, label_paths, frame_num_dict
