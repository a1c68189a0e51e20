"""Dataset for the hierarchical (multitask) CTC model on Librispeech -- mirror of
examples/librispeech/data/load_dataset_multitask_ctc.py: main task = words (`word_freq10`), sub task = characters;
same directory layout and GPU split as load_dataset_ctc.py."""
import os
import pickle
from os.path import isfile, join

import numpy as np

from tensorflow_end2end_speech_recognition_amd.utils.dataset.multitask_ctc import DatasetBase

from .load_dataset_ctc import DEFAULT_ROOTS


class Dataset(DatasetBase):

    def __init__(self, data_type, train_data_size, label_type_main, label_type_sub, batch_size, max_epoch=None,
                 splice=1, num_stack=1, num_skip=1, shuffle=False, sort_utt=False, sort_stop_epoch=None,
                 progressbar=False, num_gpu=1, is_gpu=False, dataset_root=None):
        super(Dataset, self).__init__()
        self.data_type, self.train_data_size = data_type, train_data_size
        self.label_type_main, self.label_type_sub = label_type_main, label_type_sub
        self.label_type = label_type_main
        self.batch_size = batch_size * num_gpu
        self.max_epoch = max_epoch
        self.splice, self.num_stack, self.num_skip = splice, num_stack, num_skip
        self.shuffle, self.sort_utt, self.sort_stop_epoch = shuffle, sort_utt, sort_stop_epoch
        self.progressbar = progressbar
        self.num_gpu = num_gpu
        self.is_training = data_type == 'train'
        self.is_test = 'test' in data_type
        self.padded_value = -1 if not self.is_test else None
        roots = [r for r in [dataset_root, os.environ.get('LIBRISPEECH_DATASET_ROOT')] if r] + DEFAULT_ROOTS
        for root in roots:
            input_path = join(root, 'inputs', train_data_size, data_type)
            if isfile(join(input_path, 'frame_num.pickle')):
                break
        else:
            raise IOError('frame_num.pickle not found under any of %s' % (roots,))
        with open(join(input_path, 'frame_num.pickle'), 'rb') as f:
            self.frame_num_dict = pickle.load(f)
        names = [n for n, _ in sorted(self.frame_num_dict.items(), key=lambda x: x[1 if sort_utt else 0])]

        def paths(label_type):
            base = join(root, 'labels', train_data_size, data_type, label_type)
            return np.array([join(base, n.split('-')[0], n + '.npy') for n in names])
        self.input_paths = np.array([join(input_path, n.split('-')[0], n + '.npy') for n in names])
        self.label_main_paths, self.label_sub_paths = paths(label_type_main), paths(label_type_sub)
        self.rest = set(range(len(self.input_paths)))
