"""Minibatch iterator for the CTC model -- mirror of utils/dataset/ctc.py:22-190 (DatasetBase).

Same contract: subclasses set input_paths / label_paths (.npy per utterance), batch_size, splice,
num_stack, num_skip, shuffle, sort_utt, sort_stop_epoch, num_gpu, is_test, max_epoch, rest.
__next__ -> ((inputs, labels, inputs_seq_len, input_names), is_new_epoch) with
inputs [num_gpu][B,Tmax,D*num_stack*splice] fp32 zero-padded to the GLOBAL max T (:137-139),
labels padded with -1 (:140-141), inputs_seq_len int32, split across `num_gpu` with np.array_split
(:171-182).  Sampling rules (sorted window + in-batch shuffle, random sampling, sequential) are the
reference's (:65-115).  Arrays may also be passed in memory (input_list / label_list) instead of paths."""
import random
from os.path import basename

import numpy as np

from .base import Base
from ..io.inputs.frame_stacking import stack_frame
from ..io.inputs.splicing import do_splice


class DatasetBase(Base):

    def __init__(self, *args, **kwargs):
        super(DatasetBase, self).__init__(*args, **kwargs)

    def __getitem__(self, index):
        return (np.array(self.input_paths[index]), np.array(self.label_paths[index]))

    def _load(self, paths, indices):
        out = []
        for p in (paths[i] for i in indices):
            a = np.load(p, allow_pickle=True) if isinstance(p, str) else np.asarray(p)
            out.append(a.item() if a.ndim == 0 else a)      # test-set transcripts are stored as 0-d string arrays
        return out

    # ---- the three pieces every model family's iterator shares (sampling, feature assembly, GPU split)
    def _next_indices(self, batch_size):
        """Sampling rules of :65-115: sorted window + in-batch shuffle while sort_utt, random sampling while
        shuffle, else sequential; the last partial batch closes the epoch."""
        if self.max_epoch is not None and self.epoch >= self.max_epoch:
            raise StopIteration
        if batch_size is None:
            batch_size = self.batch_size
        if self.is_new_epoch:
            self.is_new_epoch = False
        # the reference draws from the global `random`; a dataset may carry its own generator (`self.rng`, a
        # random.Random) so that data-parallel ranks keep sampling the same global batches whatever else runs
        rnd = getattr(self, 'rng', None) or random
        if self.sort_utt:
            if len(self.rest) > batch_size:
                data_indices = sorted(list(self.rest))[:batch_size]
                self.rest -= set(data_indices)
            else:
                data_indices = list(self.rest)
                self.reset()
                self.is_new_epoch = True
                self.epoch += 1
                if self.epoch == self.sort_stop_epoch:
                    self.sort_utt = False
                    self.shuffle = True
            rnd.shuffle(data_indices)
        elif self.shuffle:
            if len(self.rest) > batch_size:
                data_indices = rnd.sample(list(self.rest), batch_size)
                self.rest -= set(data_indices)
            else:
                data_indices = list(self.rest)
                self.reset()
                self.is_new_epoch = True
                self.epoch += 1
                rnd.shuffle(data_indices)
        else:
            if len(self.rest) > batch_size:
                data_indices = sorted(list(self.rest))[:batch_size]
                self.rest -= set(data_indices)
            else:
                data_indices = list(self.rest)
                self.reset()
                self.is_new_epoch = True
                self.epoch += 1
        return data_indices

    def _assemble_inputs(self, data_indices):
        """:117-166 -> (inputs [B,Tmax,D*num_stack*splice] fp32 zero-padded, inputs_seq_len int32, input_names)."""
        input_list = self._load(self.input_paths, data_indices)
        if not hasattr(self, 'input_size'):
            self.input_size = input_list[0].shape[1]
            if self.num_stack is not None and self.num_skip is not None:
                self.input_size *= self.num_stack
        device_assembly = getattr(self, 'device_assembly', False)
        if not device_assembly:
            input_list = stack_frame(input_list, self.num_stack, self.num_skip, progressbar=False)
        max_frame_num = max(map(lambda x: x.shape[0], input_list))
        # device_assembly: yield the raw padded features; utils/io/inputs/device.py assemble() stacks and
        # splices them on the GPU (same result, no Python loop over utterances x splice)
        width = input_list[0].shape[1] if device_assembly else self.input_size * self.splice
        inputs = np.zeros((len(data_indices), max_frame_num, width), dtype=np.float32)
        inputs_seq_len = np.zeros((len(data_indices),), dtype=np.int32)
        input_names = [basename(p).split('.')[0] if isinstance(p, str) else str(i)
                       for i, p in ((i, self.input_paths[i]) for i in data_indices)]
        for i_batch in range(len(data_indices)):
            data_i = np.asarray(input_list[i_batch], dtype=np.float64)
            frame_num, input_size = data_i.shape
            if not device_assembly:
                data_i = do_splice(data_i.reshape(1, frame_num, input_size), splice=self.splice, batch_size=1,
                                   num_stack=self.num_stack).reshape(frame_num, -1)
            inputs[i_batch, :frame_num, :] = data_i
            inputs_seq_len[i_batch] = frame_num
        return inputs, inputs_seq_len, np.array(input_names)

    def _split(self, a):
        """:171-182: contiguous np.array_split shards, or a leading axis of 1."""
        if self.num_gpu > 1:
            return np.array_split(a, self.num_gpu, axis=0)
        return a[np.newaxis]

    def __next__(self, batch_size=None):
        data_indices = self._next_indices(batch_size)
        self.padded_value = -1 if not self.is_test else None
        inputs, inputs_seq_len, input_names = self._assemble_inputs(data_indices)
        label_list = self._load(self.label_paths, data_indices)
        max_seq_len = max(map(len, label_list))
        labels = np.array([[self.padded_value] * max_seq_len] * len(data_indices))
        for i_batch in range(len(data_indices)):
            if self.is_test:
                labels[i_batch, 0] = label_list[i_batch]
            else:
                labels[i_batch, :len(label_list[i_batch])] = label_list[i_batch]
        self.iteration += len(data_indices)
        return (self._split(inputs), self._split(labels), self._split(inputs_seq_len),
                self._split(input_names)), self.is_new_epoch
