"""Dataset for the CTC model on Librispeech -- mirror of examples/librispeech/data/load_dataset_ctc.py:20-112.

Layout: <root>/inputs/<train_data_size>/<data_type>/{frame_num.pickle, <speaker>/<utt>.npy} and
<root>/labels/<train_data_size>/<data_type>/<label_type>/<speaker>/<utt>.npy (utterance names are
`<speaker>-<chapter>-<n>`).  batch_size is PER GPU: the iterator draws batch_size * num_gpu utterances, pads them
to one global max length and splits them with np.array_split (utils/dataset/ctc.py:171-182).  The corpus root is
`dataset_root=` / $LIBRISPEECH_DATASET_ROOT, then the reference's two site paths (:68-69)."""
import os
import pickle
from os.path import isfile, join

import numpy as np

from tensorflow_end2end_speech_recognition_amd.utils.dataset.ctc import DatasetBase

DEFAULT_ROOTS = ['/data/inaguma/librispeech', '/n/sd8/inaguma/corpus/librispeech/dataset']


class Dataset(DatasetBase):

    def __init__(self, data_type, train_data_size, label_type, batch_size, max_epoch=None, splice=1, num_stack=1,
                 num_skip=1, shuffle=False, sort_utt=False, sort_stop_epoch=None, progressbar=False, num_gpu=1,
                 dataset_root=None, device_assembly=False):
        super(Dataset, self).__init__()
        self.data_type, self.train_data_size, self.label_type = data_type, train_data_size, label_type
        self.batch_size = batch_size * num_gpu
        self.max_epoch = max_epoch
        self.splice, self.num_stack, self.num_skip = splice, num_stack, num_skip
        self.shuffle, self.sort_utt, self.sort_stop_epoch = shuffle, sort_utt, sort_stop_epoch
        self.progressbar = progressbar
        self.num_gpu = num_gpu
        self.is_test = 'test' in data_type
        self.padded_value = -1 if not self.is_test else None
        self.device_assembly = device_assembly
        roots = [r for r in [dataset_root, os.environ.get('LIBRISPEECH_DATASET_ROOT')] if r] + DEFAULT_ROOTS
        for root in roots:
            input_path = join(root, 'inputs', train_data_size, data_type)
            if isfile(join(input_path, 'frame_num.pickle')):
                break
        else:
            raise IOError('frame_num.pickle not found under any of %s (inputs/%s/%s/)' %
                          (roots, train_data_size, data_type))
        label_path = join(root, 'labels', train_data_size, data_type, label_type)
        with open(join(input_path, 'frame_num.pickle'), 'rb') as f:
            self.frame_num_dict = pickle.load(f)
        axis = 1 if sort_utt else 0
        input_paths, label_paths = [], []
        for utt_name, frame_num in sorted(self.frame_num_dict.items(), key=lambda x: x[axis]):
            speaker = utt_name.split('-')[0]
            input_paths.append(join(input_path, speaker, utt_name + '.npy'))
            label_paths.append(join(label_path, speaker, utt_name + '.npy'))
        self.input_paths = np.array(input_paths)
        self.label_paths = np.array(label_paths)
        self.rest = set(range(len(self.input_paths)))
