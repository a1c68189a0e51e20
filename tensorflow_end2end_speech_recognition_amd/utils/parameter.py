"""Parameter storage for the eager HIP models.

The reference keeps variables in the TF graph (tf.get_variable under variable
scopes; names listed in SURVEY.md Appendix C) and counts them with
utils/parameter.py:9 count_total_parameters.  Here ALL trainable variables of a
model live in ONE flat fp32 device buffer (`ParamStore.flat`) with a parallel
flat gradient buffer: per-variable tf.clip_by_norm, weight decay, the optimizer
update and the data-parallel RCCL all-reduce are then each a single pass over
one contiguous buffer (bucket) instead of one op per variable.

Checkpoint layout: `state_dict()` is a flat {reference variable name -> tensor}
map using exactly the TF names, so a reader of `model.ckpt-E` style maps finds
`blstm_hidden1/fw/lstm_cell/kernel` etc.
"""
import numpy as np
import torch

from .. import ops
from .._lib import ASR_BF16


class ParamStore(object):
    def __init__(self, device):
        self.device = device
        self._specs = []       # (name, shape, init array)
        self._index = {}
        self.flat = None
        self.grad = None
        self.views = {}
        self.gviews = {}
        self.finalized = False
        self._shadow = None
        self._shadow_dirty = True

    # -- declaration (like tf.get_variable at graph-build time)
    def declare(self, name, shape, init):
        if self.finalized:
            raise RuntimeError('ParamStore already finalized; cannot declare %s' % name)
        if name in self._index:
            raise ValueError('variable %s already exists' % name)
        init = np.asarray(init, dtype=np.float32).reshape(shape)
        self._index[name] = len(self._specs)
        self._specs.append((name, tuple(int(s) for s in shape), init))

    def finalize(self):
        """Lay the variables out back to back (each start 16-element aligned)."""
        offs = [0]
        starts = []
        pos = 0
        for name, shape, _ in self._specs:
            n = int(np.prod(shape)) if len(shape) else 1
            starts.append(pos)
            pos += (n + 15) // 16 * 16
            offs.append(pos)
        self.total = pos
        host = np.zeros(self.total, dtype=np.float32)
        for (name, shape, init), st in zip(self._specs, starts):
            host[st:st + init.size] = init.ravel()
        self.flat = torch.from_numpy(host).to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self.offsets_host = np.asarray(offs, dtype=np.int64)
        for (name, shape, init), st in zip(self._specs, starts):
            n = init.size
            self.views[name] = self.flat[st:st + n].view(shape)
            self.gviews[name] = self.grad[st:st + n].view(shape)
        self.names = [s[0] for s in self._specs]
        self.plan = ops.ClipPlan(self.offsets_host, self.flat.device)
        # weight-decay rule of models/ctc/ctc.py:283-285: every variable whose lower-cased
        # name does not contain 'bias' (peepholes and v_a ARE decayed)
        mask = np.array([0 if 'bias' in n.lower() else 1 for n in self.names], dtype=np.uint8)
        self.decay_mask = torch.from_numpy(mask).to(self.flat.device)
        self.finalized = True
        self._shadow_dirty = True

    def __getitem__(self, name):
        return self.views[name]

    def g(self, name):
        return self.gviews[name]

    def bucket(self, first, last):
        """Contiguous run of variables [first .. last] (declaration order) as a bucket of the flat buffers:
        dict(start, end, grad = view of the flat gradient buffer, plan = ClipPlan of the run).  Used by the
        data-parallel step to clip + all-reduce a layer's gradients as soon as they are complete."""
        i0, i1 = self._index[first], self._index[last]
        if i1 < i0:
            raise ValueError('bucket: %s is declared after %s' % (first, last))
        off = self.offsets_host[i0:i1 + 2] - self.offsets_host[i0]
        start, end = int(self.offsets_host[i0]), int(self.offsets_host[i1 + 1])
        return dict(start=start, end=end, names=self.names[i0:i1 + 1], grad=self.grad[start:end],
                    plan=ops.ClipPlan(off, self.flat.device))

    def num_params(self):
        return int(sum(int(np.prod(s[1])) if len(s[1]) else 1 for s in self._specs))

    def mark_dirty(self):
        self._shadow_dirty = True

    # -- bf16 shadow of the weights for the MFMA operand path: per variable, refreshed on first use after an update
    # (a step of the 5x256 CTC model touches one shadow variable, the 128 KB output matrix -- casting the whole 28 MB
    # buffer every step was 42 MB of traffic beside the first recurrence kernel)
    def shadow(self, dtype):
        if dtype != ASR_BF16:
            return self
        if self._shadow is None:
            self._shadow = _Shadow(self)
        if self._shadow_dirty:
            self._shadow.invalidate()
            self._shadow_dirty = False
        return self._shadow

    def state_dict(self):
        return {n: self.views[n].detach().clone() for n in self.names}

    def load_state_dict(self, sd):
        for n in self.names:
            self.views[n].copy_(sd[n].to(self.flat.device).view(self.views[n].shape))
        self.mark_dirty()


class _Shadow(object):
    def __init__(self, store):
        self.store = store
        self.flat = torch.empty(store.total, dtype=torch.bfloat16, device=store.flat.device)
        self.views = {}
        self.spans = {}
        for name in store.names:
            v = store.views[name]
            st = v.storage_offset()
            self.views[name] = self.flat[st:st + v.numel()].view(v.shape)
            self.spans[name] = (st, st + v.numel())
        self.fresh = set()

    def invalidate(self):
        self.fresh = set()

    def __getitem__(self, name):
        if name not in self.fresh:      # cast this variable on the current stream (its consumers follow on it)
            a, b = self.spans[name]
            ops.cast_from_f32(self.store.flat[a:b], ASR_BF16, out=self.flat[a:b])
            self.fresh.add(name)
        return self.views[name]


def count_total_parameters(variables):
    """utils/parameter.py:9 -- returns (dict name->count, total)."""
    d = {}
    total = 0
    for name, v in variables.items():
        n = int(v.numel())
        d[name] = n
        total += n
    return d, total
