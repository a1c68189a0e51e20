"""Minibatch iterator for the joint CTC-attention model -- mirror of utils/dataset/joint_ctc_attention.py:22-209.

One target per utterance, emitted twice (:135-180): att_labels = <SOS> y <EOS> padded with <EOS>, ctc_labels = y
padded with -1:
__next__ -> ((inputs, att_labels, ctc_labels, inputs_seq_len, att_labels_seq_len, input_names), is_new_epoch)."""
import numpy as np

from .ctc import DatasetBase as _CTCDatasetBase


class DatasetBase(_CTCDatasetBase):

    def __next__(self, batch_size=None):
        data_indices = self._next_indices(batch_size)
        self.att_padded_value = self.eos_index if not self.is_test else None
        self.ctc_padded_value = -1 if not self.is_test else None
        inputs, inputs_seq_len, input_names = self._assemble_inputs(data_indices)
        label_list = self._load(self.label_paths, data_indices)
        max_seq_len = max(map(len, label_list))
        att_labels = np.array([[self.att_padded_value] * (max_seq_len + 2)] * len(data_indices))
        ctc_labels = np.array([[self.ctc_padded_value] * max_seq_len] * len(data_indices))
        att_labels_seq_len = np.zeros((len(data_indices),), dtype=np.int32)
        for i_batch in range(len(data_indices)):
            if self.is_test:
                att_labels[i_batch, 0] = label_list[i_batch]
                ctc_labels[i_batch, 0] = label_list[i_batch]
            else:
                n = len(label_list[i_batch])
                att_labels[i_batch, 0] = self.sos_index
                att_labels[i_batch, 1:n + 1] = label_list[i_batch]
                att_labels[i_batch, n + 1] = self.eos_index
                ctc_labels[i_batch, :n] = label_list[i_batch]
            att_labels_seq_len[i_batch] = len(label_list[i_batch]) + 2
        self.iteration += len(data_indices)
        return (self._split(inputs), self._split(att_labels), self._split(ctc_labels), self._split(inputs_seq_len),
                self._split(att_labels_seq_len), self._split(input_names)), self.is_new_epoch
