"""Loader of tests/golden/tfshim_v1.npz: what the REFERENCE'S OWN model code computed when it was executed on the eager
TensorFlow stand-in (generator: tests/golden/make_golden_tfshim.py, which needs /root/reference; the fixture travels).

A case = one run of reference code.  Variables are not stored: each one is uniform(-scale, scale) from its own seed
(`var_values`, the generator's recipe verbatim), the meta block holds name -> [shape, seed, scale].  Tensors above
20 000 elements (the hard-coded 64 / 128-channel VGG filters) are stored as every 97th element + [sum, sum of squares]."""
import json
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tfshim_v1.npz')
STRIDE = 97
_cache = {}


def var_values(shape, seed, scale):
    return np.random.RandomState(seed).uniform(-scale, scale, size=tuple(shape))


def load():
    if 'z' not in _cache:
        z = np.load(PATH)
        _cache['z'] = {k: z[k] for k in z.files}
        _cache['meta'] = json.loads(bytes(_cache['z']['meta_json']).decode())
    return _cache['z'], _cache['meta']


def cases(kind):
    _, meta = load()
    return sorted(c for c, m in meta.items() if m['kind'] == kind)


class Case(object):
    def __init__(self, name):
        z, meta = load()
        self.name, self.meta = name, meta[name]
        self._g = {}
        pre = name + '|'
        for k, v in z.items():
            if k.startswith(pre):
                _, group, nm = k.split('|')
                self._g.setdefault(group, {})[nm] = v

    def group(self, g):
        return self._g.get(g, {})

    def variables(self):
        return {n: var_values(*rec) for n, rec in self.meta['vars'].items()}

    def check_grads(self, got, tol, group='grad', skip=()):
        """got: {variable name: array}.  Every gradient the reference's graph produced must be matched; variables whose
        gradient is None there (meta none_grads) must come out as zeros / absent."""
        worst = 0.0
        names = set(self.group(group)) | set(self.group(group + '_sub'))
        assert names or not self.meta['vars'], self.name
        for n in sorted(names):
            if n in skip:
                continue
            g = np.asarray(got[n], dtype=np.float64)
            if n in self.group(group):
                ref = self.group(group)[n]
                assert g.shape == ref.shape, (n, g.shape, ref.shape)
                err = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-6)
            else:
                sub, stat = self.group(group + '_sub')[n], self.group(group + '_stat')[n]
                scale = max(np.abs(sub).max(), 1e-6)
                err = max(np.abs(g.reshape(-1)[::STRIDE] - sub).max() / scale,
                          abs(g.sum() - stat[0]) / max(abs(stat[0]), scale),
                          abs((g * g).sum() - stat[1]) / max(stat[1], 1e-30))
            assert err < tol, (self.name, n, err)
            worst = max(worst, err)
        for n in self.meta.get('none_grads', []):
            if n in got and got[n] is not None:
                assert np.abs(np.asarray(got[n])).max() == 0.0, (self.name, n)
        return worst
