"""Dataset for the joint CTC-attention model on TIMIT -- mirror of
examples/timit/data/load_dataset_joint_ctc_attention.py:20-100."""
from tensorflow_end2end_speech_recognition_amd.utils.dataset.joint_ctc_attention import DatasetBase

from ._paths import utterance_paths


class Dataset(DatasetBase):

    def __init__(self, data_type, label_type, batch_size, map_file_path, max_epoch=None, splice=1, num_stack=1,
                 num_skip=1, shuffle=False, sort_utt=False, sort_stop_epoch=None, progressbar=False,
                 dataset_root=None):
        super(Dataset, self).__init__(map_file_path=map_file_path)
        self.is_test = data_type == 'test'
        self.data_type, self.label_type = data_type, label_type
        self.batch_size, self.max_epoch = batch_size, max_epoch
        self.splice, self.num_stack, self.num_skip = splice, num_stack, num_skip
        self.shuffle, self.sort_utt, self.sort_stop_epoch = shuffle, sort_utt, sort_stop_epoch
        self.progressbar = progressbar
        self.num_gpu = 1
        self.input_paths, (self.label_paths,), self.frame_num_dict = utterance_paths(
            data_type, [label_type], sort_utt, dataset_root)
        self.rest = set(range(len(self.input_paths)))
