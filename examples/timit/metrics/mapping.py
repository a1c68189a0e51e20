"""61 / 48 -> 39 phone folding -- mirror of examples/timit/metrics/mapping.py:11-53 (Map2phone39).

With `map_file_path` the three-column file of the reference is read (`<phone61> <phone48|nan> <phone39>`,
examples/timit/metrics/mapping_files/phone2phone.txt).  Without it the standard Lee & Hon (1989) folding
below is used; tests/test_host_io.py checks it phone by phone against what the reference's class returns on
its own file (tests/golden/labels_v1.json)."""

# phone61: (phone48 or None when the phone is dropped, phone39)
_FOLD = {
    'aa': ('aa', 'aa'), 'ae': ('ae', 'ae'), 'ah': ('ah', 'ah'), 'ao': ('ao', 'aa'), 'aw': ('aw', 'aw'),
    'ax': ('ax', 'ah'), 'ax-h': ('ax', 'ah'), 'axr': ('er', 'er'), 'ay': ('ay', 'ay'), 'b': ('b', 'b'),
    'bcl': ('vcl', 'sil'), 'ch': ('ch', 'ch'), 'd': ('d', 'd'), 'dcl': ('vcl', 'sil'), 'dh': ('dh', 'dh'),
    'dx': ('dx', 'dx'), 'eh': ('eh', 'eh'), 'el': ('el', 'l'), 'em': ('m', 'm'), 'en': ('en', 'n'),
    'eng': ('ng', 'ng'), 'epi': ('epi', 'sil'), 'er': ('er', 'er'), 'ey': ('ey', 'ey'), 'f': ('f', 'f'),
    'g': ('g', 'g'), 'gcl': ('vcl', 'sil'), 'h#': ('sil', 'sil'), 'hh': ('hh', 'hh'), 'hv': ('hh', 'hh'),
    'ih': ('ih', 'ih'), 'ix': ('ix', 'ih'), 'iy': ('iy', 'iy'), 'jh': ('jh', 'jh'), 'k': ('k', 'k'),
    'kcl': ('cl', 'sil'), 'l': ('l', 'l'), 'm': ('m', 'm'), 'n': ('n', 'n'), 'ng': ('ng', 'ng'),
    'nx': ('n', 'n'), 'ow': ('ow', 'ow'), 'oy': ('oy', 'oy'), 'p': ('p', 'p'), 'pau': ('sil', 'sil'),
    'pcl': ('cl', 'sil'), 'q': (None, ''), 'r': ('r', 'r'), 's': ('s', 's'), 'sh': ('sh', 'sh'),
    't': ('t', 't'), 'tcl': ('cl', 'sil'), 'th': ('th', 'th'), 'uh': ('uh', 'uh'), 'uw': ('uw', 'uw'),
    'ux': ('uw', 'uw'), 'v': ('v', 'v'), 'w': ('w', 'w'), 'y': ('y', 'y'), 'z': ('z', 'z'), 'zh': ('zh', 'sh'),
}


class Map2phone39(object):
    def __init__(self, label_type, map_file_path=None):
        self.label_type = label_type
        self.map_dict = {}
        rows = []
        if map_file_path is not None:
            with open(map_file_path) as f:
                rows = [line.strip().split() for line in f if line.strip()]
        else:
            rows = [[p61, p48 if p48 is not None else 'nan', p39] for p61, (p48, p39) in _FOLD.items()]
        for r in rows:
            if label_type == 'phone61':
                self.map_dict[r[0]] = r[2] if r[1] != 'nan' else ''
            elif label_type == 'phone48' and r[1] != 'nan':
                self.map_dict[r[1]] = r[2]

    def __call__(self, phone_list):
        """list of phone strings -> list of 39-set phone strings ('q' of the 61 set is dropped)."""
        if self.label_type == 'phone39':
            return phone_list
        return [p for p in (self.map_dict[ph] for ph in phone_list) if p != '']
