"""Minibatch iterator for the attention model -- mirror of utils/dataset/attention.py:22-203 (DatasetBase).

As the CTC iterator (same sampling, stacking / splicing, zero padding to the global max T, GPU split), with the
label side of :136-175: every target becomes <SOS> y <EOS>, rows are padded with <EOS>, and labels_seq_len =
len(y) + 2 is returned as well:
__next__ -> ((inputs, labels, inputs_seq_len, labels_seq_len, input_names), is_new_epoch).
sos_index / eos_index come from the subclass' map file ('<' and '>', utils/dataset/base.py:36-42)."""
import numpy as np

from .ctc import DatasetBase as _CTCDatasetBase


class DatasetBase(_CTCDatasetBase):

    def __next__(self, batch_size=None):
        data_indices = self._next_indices(batch_size)
        self.padded_value = self.eos_index if not self.is_test else None
        inputs, inputs_seq_len, input_names = self._assemble_inputs(data_indices)
        label_list = self._load(self.label_paths, data_indices)
        max_seq_len = max(map(len, label_list)) + 2                       # + <SOS> and <EOS>
        labels = np.array([[self.padded_value] * max_seq_len] * len(data_indices))
        labels_seq_len = np.zeros((len(data_indices),), dtype=np.int32)
        for i_batch in range(len(data_indices)):
            if self.is_test:
                labels[i_batch, 0] = label_list[i_batch]                  # the transcript is kept as a string
            else:
                n = len(label_list[i_batch])
                labels[i_batch, 0] = self.sos_index
                labels[i_batch, 1:n + 1] = label_list[i_batch]
                labels[i_batch, n + 1] = self.eos_index
            labels_seq_len[i_batch] = len(label_list[i_batch]) + 2
        self.iteration += len(data_indices)
        return (self._split(inputs), self._split(labels), self._split(inputs_seq_len), self._split(labels_seq_len),
                self._split(input_names)), self.is_new_epoch
