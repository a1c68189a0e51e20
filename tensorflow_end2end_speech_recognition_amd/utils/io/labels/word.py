"""Index -> word map -- mirror of utils/io/labels/word.py:12-42 (Idx2word)."""
import numpy as np

from .phone import _read_map


class Idx2word(object):
    def __init__(self, map_file_path):
        self.map_dict = dict((i, w) for w, i in _read_map(map_file_path))

    def __call__(self, index_list, padded_value=-1):
        assert type(index_list) == np.ndarray, 'index_list should be np.ndarray.'
        return [self.map_dict[int(i)] for i in index_list if i != -1]      # -1, as the reference (:36)
