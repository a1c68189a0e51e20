"""Minibatch iterator for the multitask CTC model -- mirror of utils/dataset/multitask_ctc.py:22-206.

Subclasses set label_main_paths / label_sub_paths instead of label_paths; both label sets are padded with -1
(:147-177; only the main labels are strings on the test set):
__next__ -> ((inputs, labels_main, labels_sub, inputs_seq_len, input_names), is_new_epoch)."""
import numpy as np

from .ctc import DatasetBase as _CTCDatasetBase


class DatasetBase(_CTCDatasetBase):

    def __getitem__(self, index):
        return (np.array(self.input_paths[index]), np.array(self.label_main_paths[index]),
                np.array(self.label_sub_paths[index]))

    def __next__(self, batch_size=None):
        data_indices = self._next_indices(batch_size)
        self.padded_value = -1 if not self.is_test else None
        inputs, inputs_seq_len, input_names = self._assemble_inputs(data_indices)
        main_list = self._load(self.label_main_paths, data_indices)
        sub_list = self._load(self.label_sub_paths, data_indices)
        labels_main = np.array([[self.padded_value] * max(map(len, main_list))] * len(data_indices))
        labels_sub = np.array([[self.padded_value] * max(map(len, sub_list))] * len(data_indices))
        for i_batch in range(len(data_indices)):
            if self.is_test:
                labels_main[i_batch, 0] = main_list[i_batch]
            else:
                labels_main[i_batch, :len(main_list[i_batch])] = main_list[i_batch]
            labels_sub[i_batch, :len(sub_list[i_batch])] = sub_list[i_batch]
        self.iteration += len(data_indices)
        return (self._split(inputs), self._split(labels_main), self._split(labels_sub), self._split(inputs_seq_len),
                self._split(input_names)), self.is_new_epoch
