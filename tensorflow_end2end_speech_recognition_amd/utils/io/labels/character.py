"""Character <-> index maps -- mirror of utils/io/labels/character.py:11-121 (Char2idx, Idx2char).
Map file: `<token> <index>` per line; tokens may be double letters (character_capital_divide.txt)."""
import numpy as np

from .phone import _read_map


class Char2idx(object):
    def __init__(self, map_file_path, double_letter=False):
        self.double_letter = double_letter
        self.map_dict = dict(_read_map(map_file_path))

    def __call__(self, str_char):
        """string -> list of indices.  double_letter: scanning left to right, two adjacent characters that
        form a token of the map are emitted as that token (greedy, non-overlapping; :38-62)."""
        chars = list(str_char)
        if not self.double_letter:
            return [self.map_dict[c] for c in chars]
        out = []
        i, n = 0, len(chars)
        while i < n:
            if i + 1 < n and chars[i] + chars[i + 1] in self.map_dict:
                out.append(self.map_dict[chars[i] + chars[i + 1]])
                i += 2
            else:
                out.append(self.map_dict[chars[i]])
                i += 1
        return out


class Idx2char(object):
    def __init__(self, map_file_path, capital_divide=False, space_mark=' '):
        self.capital_divide = capital_divide
        self.space_mark = space_mark
        self.map_dict = dict((i, c) for c, i in _read_map(map_file_path))

    def __call__(self, index_list, padded_value=-1):
        """np.ndarray of indices -> string.  capital_divide (:103-114): a token that compares inside
        ['A', 'Z'] and is not the first one starts a new word (space_mark + lower-cased token); every
        other token is lower-cased."""
        assert type(index_list) == np.ndarray, 'index_list should be np.ndarray.'
        toks = [self.map_dict[int(i)] for i in index_list if i != padded_value]
        if not self.capital_divide:
            return ''.join(toks)
        out = []
        for k, t in enumerate(toks):
            if k != 0 and 'A' <= t <= 'Z':
                out.append(self.space_mark + t.lower())
            else:
                out.append(t.lower())
        return ''.join(out)
