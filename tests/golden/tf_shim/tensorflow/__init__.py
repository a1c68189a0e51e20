"""A small EAGER stand-in for the parts of TensorFlow 1.x that the reference's model code calls.

TEST INFRASTRUCTURE, used by ONE script only: tests/golden/make_golden_tfshim.py puts this directory in front of
sys.path so that `import tensorflow as tf` inside the UNCHANGED files of /root/reference resolves to this module, runs
the reference's own Python (models/attention/decoders/attention_layer.py, attention_decoder.py, dynamic_decoder.py,
models/attention/bridge.py, attention_seq2seq.py, joint_ctc_attention.py, models/ctc/ctc.py,
models/encoders/core/{blstm,lstm,vgg_blstm,cnn_util}.py, models/recurrent/layers/lstm.py, models/model_base.py) and
records what it computes as fixtures (tests/golden/tfshim_v1.npz).  Nothing in the product, the oracle or the tests
imports it; the fixtures are what travels.

What is the REFERENCE's arithmetic and what is this file's:
  * every line of the files above executes as written: which tensor meets which weight, bias placement, the float32.min
    mask and its interaction with the sharpening factor, sigmoid smoothing, input feeding, impute_finished, the bridge's
    flatten order, loss composition and weights, per-variable clipping -- all of that is the reference speaking;
  * the ops it bottoms out in are restated here from TensorFlow 1.3's documented semantics on
    torch float64 tensors (so gradients of the reference's own forward graph come from autograd): matmul / split /
    concat / where / sequence_mask / softmax / conv1d, conv2d, max_pool with SAME padding / fully_connected /
    _linear / variable scopes and names / LSTMBlockCell / GRUCell / MultiRNNCell / DropoutWrapper(keep = 1) /
    dynamic_rnn + bidirectional_dynamic_rnn (zero output and state copy-through past sequence_length,
    reverse_sequence) / TrainingHelper, GreedyEmbeddingHelper, CustomHelper / sequence_loss / while_loop and
    TensorArray (eager) / nest / clip_by_norm / the optimizers of tf.train;
  * tf.nn.ctc_loss is torch.nn.functional.ctc_loss (float64, blank = C - 1): a third implementation, independent
    of oracle/ctc.py and of the HIP kernel.

Eager execution differs from a TF1 graph in ONE way that matters to the reference: a
`tf.while_loop` body is traced once, so Python attribute updates inside `AttentionDecoder.step` (`self.attention_weights
= ...`) do not carry from iteration to iteration in the reference's graph; here the body really runs per step and they
do.  The generator records both: the eager run (= the intended recurrence, `prev_alpha='carry'`) and a run in which the
attribute is put back to the tensor `initialize()` made before every step (= the graph the reference builds,
`prev_alpha='zeros'`).

All floating point tensors are float64 whatever dtype the caller names; integer / bool tensors keep their kind.
"""
import collections
import contextlib
import importlib.abc
import importlib.machinery
import math
import sys
import types

import numpy as np
import torch

__version__ = '1.3.0'          # the reference branches on this for `clip_cell` (models/encoders/core/blstm.py:286)

_F = torch.float64
_py_range, _py_slice, _py_bool, _py_abs = range, slice, bool, abs      # the tf.* functions below shadow these names


# ----------------------------------------------------------------------------------------------------------------
# dtypes, shapes
class DType(object):
    def __init__(self, name, torch_dtype, np_dtype):
        self.name, self._torch, self.as_numpy_dtype = name, torch_dtype, np_dtype

    @property
    def min(self):
        return float(np.finfo(self.as_numpy_dtype).min) if self.is_floating else int(np.iinfo(self.as_numpy_dtype).min)

    @property
    def max(self):
        return float(np.finfo(self.as_numpy_dtype).max) if self.is_floating else int(np.iinfo(self.as_numpy_dtype).max)

    @property
    def is_floating(self):
        return self.name.startswith('float')

    @property
    def base_dtype(self):
        return self

    def __repr__(self):
        return 'tf.' + self.name


float32 = DType('float32', _F, np.float32)
float64 = DType('float64', _F, np.float64)
float16 = DType('float16', _F, np.float16)
int32 = DType('int32', torch.int64, np.int32)
int64 = DType('int64', torch.int64, np.int64)
bool = DType('bool', torch.bool, np.bool_)          # noqa: A001  (tf.bool)


def _dtype_of(t):
    if t.dtype == torch.bool:
        return bool
    if t.dtype.is_floating_point:
        return float32
    return int32


class Dimension(object):
    def __init__(self, value):
        self.value = None if value is None else int(value.value if isinstance(value, Dimension) else value)

    def __int__(self):
        return self.value

    __index__ = __int__

    def __eq__(self, other):
        return self.value == (other.value if isinstance(other, Dimension) else other)

    def __hash__(self):
        return hash(self.value)

    def __mul__(self, other):
        return Dimension(self.value * int(other))

    __rmul__ = __mul__

    def __repr__(self):
        return 'Dimension(%s)' % self.value


class TensorShape(object):
    def __init__(self, dims):
        if dims is None:
            self._dims = None
        elif isinstance(dims, TensorShape):
            self._dims = None if dims._dims is None else list(dims._dims)
        else:
            try:
                self._dims = [Dimension(d) for d in dims]
            except TypeError:                 # TF: "treat as a singleton dimension"
                self._dims = [Dimension(dims)]

    @property
    def ndims(self):
        return None if self._dims is None else len(self._dims)

    @property
    def dims(self):
        return self._dims

    def as_list(self):
        return [d.value for d in self._dims]

    def with_rank(self, rank):
        assert self.ndims == rank, (self, rank)
        return self

    def with_rank_at_least(self, rank):
        assert self.ndims >= rank
        return self

    def concatenate(self, other):
        other = TensorShape(other)
        if self._dims is None or other._dims is None:
            return TensorShape(None)
        return TensorShape(self._dims + other._dims)

    def is_compatible_with(self, other):
        other = TensorShape(other)
        if self._dims is None or other._dims is None:
            return True
        return len(self._dims) == len(other._dims) and all(
            a.value is None or b.value is None or a.value == b.value for a, b in zip(self._dims, other._dims))

    def is_fully_defined(self):
        return self._dims is not None and all(d.value is not None for d in self._dims)

    def __getitem__(self, k):
        if isinstance(k, _py_slice):
            return TensorShape(self._dims[k])
        return self._dims[k]

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __eq__(self, other):
        try:
            return self.as_list() == TensorShape(other).as_list()
        except Exception:
            return False

    def __repr__(self):
        return 'TensorShape(%s)' % (None if self._dims is None else self.as_list())


# ----------------------------------------------------------------------------------------------------------------
# tensors
def _raw(x, like=None):
    """anything -> torch tensor (float64 / int64 / bool)."""
    if isinstance(x, Tensor):
        return x._t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, (Dimension,)):
        return torch.tensor(x.value)
    if isinstance(x, TensorShape):
        return torch.tensor(x.as_list(), dtype=torch.int64)
    if isinstance(x, (list, tuple)):
        if any(isinstance(e, (Tensor, torch.Tensor, Dimension)) for e in x):
            return torch.stack([_raw(e) for e in x]) if len(x) else torch.zeros(0, dtype=torch.int64)
        a = np.asarray(x)
    else:
        a = np.asarray(x)
    if a.dtype == np.bool_:
        return torch.from_numpy(a.copy())
    if np.issubdtype(a.dtype, np.floating):
        return torch.from_numpy(a.astype(np.float64))
    return torch.from_numpy(a.astype(np.int64))


def _ints(x):
    """shape-like (python ints, scalar Tensors, Dimensions, 1-D Tensor) -> list of python ints."""
    if isinstance(x, Tensor):
        return [int(v) for v in x._t.reshape(-1).tolist()]
    if isinstance(x, TensorShape):
        return x.as_list()
    if isinstance(x, (int, np.integer, Dimension)):
        return [int(x)]
    return [int(e._t) if isinstance(e, Tensor) else int(e) for e in x]


class Tensor(object):
    __array_priority__ = 1000

    def __init__(self, t, name=None):
        self._t = t
        self.name = name

    # --- shape / dtype surface
    @property
    def shape(self):
        return TensorShape(list(self._t.shape))

    def get_shape(self):
        return self.shape

    def set_shape(self, s):
        assert self.shape.is_compatible_with(s), (self.shape, s)

    @property
    def dtype(self):
        return _dtype_of(self._t)

    @property
    def op(self):
        return self

    @property
    def device(self):
        return ''

    def numpy(self):
        return self._t.detach().numpy()

    def eval(self):
        return self.numpy()

    # --- arithmetic
    def _bin(self, other, fn, reverse=False):
        o = _raw(other)
        a, b = (o, self._t) if reverse else (self._t, o)
        if a.dtype != b.dtype and (a.dtype.is_floating_point or b.dtype.is_floating_point):
            a, b = a.to(_F), b.to(_F)
        return Tensor(fn(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a.to(_F) / b.to(_F))
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: a.to(_F) / b.to(_F), True)
    def __floordiv__(self, o): return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode='floor'))
    def __neg__(self): return Tensor(-self._t)
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __invert__(self): return Tensor(~self._t)
    def __and__(self, o): return self._bin(o, torch.logical_and)
    def __or__(self, o): return self._bin(o, torch.logical_or)
    __hash__ = object.__hash__

    def __bool__(self):
        return _py_bool(self._t)

    def __int__(self):
        return int(self._t)

    __index__ = __int__

    def __float__(self):
        return float(self._t)

    def __getitem__(self, k):
        def conv(e):
            if isinstance(e, Tensor):
                return e._t if e._t.dim() else int(e._t)
            if isinstance(e, _py_slice):
                return _py_slice(conv(e.start) if e.start is not None else None,
                             conv(e.stop) if e.stop is not None else None, e.step)
            return e
        if isinstance(k, tuple):
            return Tensor(self._t[tuple(conv(e) for e in k)])
        return Tensor(self._t[conv(k)])

    def __iter__(self):
        for i in _py_range(self._t.shape[0]):
            yield Tensor(self._t[i])

    def __len__(self):
        return self._t.shape[0]

    def __repr__(self):
        return 'shim.Tensor(%s, shape=%s)' % (self.name, list(self._t.shape))


class Variable(Tensor):
    """tf.Variable / tf.get_variable result: a named float64 leaf that autograd differentiates."""

    def __init__(self, initial_value=None, trainable=True, name=None, dtype=None, **kw):
        full = _scoped(name or 'Variable')
        if full in _G.store:                 # eager re-execution of a traced body: the graph's one variable
            v = _G.store[full]
            self._t, self.name, self.trainable = v._t, v.name, v.trainable
            return
        t = _raw(initial_value() if callable(initial_value) else initial_value).clone()
        if t.dtype.is_floating_point:
            t = t.to(_F).requires_grad_(trainable)
        Tensor.__init__(self, t, full + ':0')
        self.trainable = trainable
        _G.store[full] = self

    def initialized_value(self):
        return self

    def assign(self, value):
        with torch.no_grad():
            self._t.copy_(_raw(value))
        return self


# ----------------------------------------------------------------------------------------------------------------
# graph state: variable store, scopes, collections, rng
class _Graph(object):
    def __init__(self):
        self.reset()
        self.rng = np.random.RandomState(0)

    def reset(self, keep_variables=False):
        if not keep_variables:
            self.store = collections.OrderedDict()
        self.collections = collections.defaultdict(list)
        self.scopes = [VariableScope('', None, None)]


class VariableScope(object):
    def __init__(self, name, initializer, reuse):
        self.name, self.initializer, self.reuse = name, initializer, reuse
        self.caching_device = None
        self.original_name_scope = name + '/' if name else ''

    def set_caching_device(self, d):
        self.caching_device = d

    def set_partitioner(self, p):
        pass

    def reuse_variables(self):
        self.reuse = True


_G = None


def _scoped(name):
    cur = _G.scopes[-1].name
    return (cur + '/' + name) if cur else name


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, values=None, initializer=None, reuse=None, **kw):
    cur = _G.scopes[-1]
    if isinstance(name_or_scope, VariableScope):
        new = VariableScope(name_or_scope.name, initializer or name_or_scope.initializer or cur.initializer,
                            reuse if reuse is not None else (name_or_scope.reuse or cur.reuse))
    else:
        # models/recurrent/layers/lstm.py:143 passes the cell object first and the name second
        nm = name_or_scope if isinstance(name_or_scope, str) else default_name
        assert isinstance(nm, str), (name_or_scope, default_name)
        new = VariableScope(_scoped(nm), initializer or cur.initializer, reuse if reuse is not None else cur.reuse)
    _G.scopes.append(new)
    try:
        yield new
    finally:
        _G.scopes.pop()


def get_variable_scope():
    return _G.scopes[-1]


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield name


@contextlib.contextmanager
def control_dependencies(deps):
    yield


@contextlib.contextmanager
def device(d):
    yield


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **kw):
    full = _scoped(name)
    if full in _G.store:
        v = _G.store[full]
        if shape is not None:
            assert list(v._t.shape) == _ints(shape), (full, list(v._t.shape), _ints(shape))
        return v
    init = initializer or _G.scopes[-1].initializer or glorot_uniform_initializer()
    shp = _ints(shape) if shape is not None else []
    if isinstance(init, (Tensor, np.ndarray, float, int)):
        val = _raw(init)
    else:
        val = _raw(init(shp))
    out = object.__new__(Variable)
    Tensor.__init__(out, val.to(_F).clone().requires_grad_(trainable), full + ':0')
    out.trainable = trainable
    _G.store[full] = out
    return out


def trainable_variables():
    return [v for v in _G.store.values() if v.trainable]


def global_variables():
    return list(_G.store.values())


def add_to_collection(name, value):
    _G.collections[name].append(value)


def get_collection(name, scope=None):
    return list(_G.collections[name])


def reset_default_graph():
    _G.reset()


class GraphKeys(object):
    UPDATE_OPS = 'update_ops'
    TRAINABLE_VARIABLES = 'trainable_variables'
    GLOBAL_VARIABLES = 'variables'


# shim-only helpers for the generator ------------------------------------------------------------------------
def shim_reset(keep_variables=False, seed=None):
    _G.reset(keep_variables)
    if seed is not None:
        _G.rng = np.random.RandomState(seed)


def shim_variables():
    return _G.store


def shim_set_variable(name, value):
    v = _G.store[name]
    with torch.no_grad():
        v._t.copy_(torch.as_tensor(np.asarray(value), dtype=_F))


def shim_zero_grads():
    for v in _G.store.values():
        v._t.grad = None


# ----------------------------------------------------------------------------------------------------------------
# initializers / random
def random_uniform_initializer(minval=0.0, maxval=1.0, seed=None, dtype=None):
    return lambda shape, dtype=None, partition_info=None: _G.rng.uniform(minval, maxval, size=_ints(shape))


def _trunc_normal(shape, stddev, mean=0.0):
    out = _G.rng.normal(size=_ints(shape))
    bad = np.abs(out) > 2
    while bad.any():
        out[bad] = _G.rng.normal(size=int(bad.sum()))
        bad = np.abs(out) > 2
    return out * stddev + mean


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return lambda shape, dtype=None, partition_info=None: _trunc_normal(shape, stddev, mean)


def zeros_initializer(dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.zeros(_ints(shape))


def ones_initializer(dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.ones(_ints(shape))


def constant_initializer(value=0.0, dtype=None):
    return lambda shape, dtype=None, partition_info=None: np.full(_ints(shape), float(value))


def glorot_uniform_initializer(seed=None, dtype=None):
    def f(shape, dtype=None, partition_info=None):
        s = _ints(shape)
        fan = (s[0] + s[-1]) if len(s) >= 2 else max(sum(s), 1)
        lim = math.sqrt(6.0 / fan)
        return _G.rng.uniform(-lim, lim, size=s)
    return f


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return Tensor(_raw(_trunc_normal(shape, stddev, mean)))


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    return Tensor(_raw(_G.rng.normal(mean, stddev, size=_ints(shape))))


def random_uniform(shape, minval=0, maxval=1.0, dtype=None, seed=None, name=None):
    return Tensor(_raw(_G.rng.uniform(minval, maxval, size=_ints(shape))))


# ----------------------------------------------------------------------------------------------------------------
# array / math ops
def _tdtype(dtype, default=_F):
    return default if dtype is None else dtype._torch


def convert_to_tensor(value, dtype=None, name=None, **kw):
    t = _raw(value)
    if dtype is not None and not isinstance(value, Tensor):
        t = t.to(dtype._torch)
    return value if isinstance(value, Tensor) and dtype is None else Tensor(t)


def constant(value, dtype=None, shape=None, name=None):
    t = _raw(value)
    if dtype is not None:
        t = t.to(dtype._torch)
    if shape is not None:
        t = t.expand(_ints(shape)).clone() if t.dim() == 0 else t.reshape(_ints(shape))
    return Tensor(t)


def identity(x, name=None):
    return convert_to_tensor(x)


def zeros(shape, dtype=None, name=None):
    return Tensor(torch.zeros(_ints(shape), dtype=_tdtype(dtype)))


def ones(shape, dtype=None, name=None):
    return Tensor(torch.ones(_ints(shape), dtype=_tdtype(dtype)))


def zeros_like(x, dtype=None, name=None):
    return Tensor(torch.zeros_like(_raw(x)))


def ones_like(x, dtype=None, name=None):
    return Tensor(torch.ones_like(_raw(x)))


def fill(dims, value, name=None):
    return Tensor(torch.full(_ints(dims), value, dtype=torch.int64 if isinstance(value, (int, np.integer)) else _F))


def shape(x, name=None, out_type=None):
    return Tensor(torch.tensor(list(_raw(x).shape), dtype=torch.int64))


def size(x, name=None):
    return Tensor(torch.tensor(_raw(x).numel()))


def rank(x, name=None):
    return Tensor(torch.tensor(_raw(x).dim()))


def range(*args, **kw):        # noqa: A001  (tf.range)
    vals = [int(_raw(a)) for a in args]
    return Tensor(torch.arange(*vals))


def cast(x, dtype, name=None):
    return Tensor(_raw(x).to(dtype._torch))


def to_int32(x, name=None):
    return cast(x, int32)


def to_float(x, name=None):
    return cast(x, float32)


def to_int64(x, name=None):
    return cast(x, int64)


def reshape(x, shape, name=None):
    return Tensor(_raw(x).reshape(_ints(shape)))


def transpose(x, perm=None, name=None):
    t = _raw(x)
    if perm is None:
        perm = list(reversed(list(_py_range(t.dim()))))
    return Tensor(t.permute(*_ints(perm)))


def expand_dims(x, axis=None, name=None, dim=None):
    return Tensor(_raw(x).unsqueeze(axis if axis is not None else dim))


def squeeze(x, axis=None, name=None, squeeze_dims=None):
    axis = axis if axis is not None else squeeze_dims
    t = _raw(x)
    if axis is None:
        return Tensor(t.squeeze())
    for a in sorted(_ints(axis), reverse=True):
        t = t.squeeze(a)
    return Tensor(t)


def concat(values, axis, name=None):
    if isinstance(values, (int, np.integer)) and not isinstance(axis, (int, np.integer)):   # tf 0.x argument order
        values, axis = axis, values
    ts = [_raw(v) for v in values]
    ts = [t.reshape(1) if t.dim() == 0 else t for t in ts]
    if any(t.dtype.is_floating_point for t in ts):
        ts = [t.to(_F) for t in ts]
    return Tensor(torch.cat(ts, dim=int(axis)))


def stack(values, axis=0, name=None):
    return Tensor(torch.stack([_raw(v) for v in values], dim=axis))


def unstack(value, num=None, axis=0, name=None):
    return [Tensor(t) for t in torch.unbind(_raw(value), dim=axis)]


def split(value, num_or_size_splits, axis=0, num=None, name=None):
    if isinstance(value, (int, np.integer)) and not isinstance(num_or_size_splits, (int, np.integer, list, tuple)):
        axis, num_or_size_splits, value = value, axis, num_or_size_splits          # tf 0.x argument order
    t = _raw(value)
    if isinstance(num_or_size_splits, (int, np.integer)):
        sizes = [t.shape[axis] // num_or_size_splits] * num_or_size_splits
    else:
        sizes = _ints(num_or_size_splits)
    return [Tensor(p) for p in torch.split(t, sizes, dim=axis)]


def tile(x, multiples, name=None):
    return Tensor(_raw(x).repeat(*_ints(multiples)))


def slice(x, begin, size, name=None):      # noqa: A001  (tf.slice)
    t = _raw(x)
    idx = []
    for d, (b, s) in enumerate(zip(_ints(begin), _ints(size))):
        idx.append(_py_slice(b, t.shape[d] if s == -1 else b + s))
    return Tensor(t[tuple(idx)])


def where(condition, x=None, y=None, name=None):
    c = _raw(condition)
    if x is None:
        return Tensor(torch.nonzero(c))
    a, b = _raw(x), _raw(y)
    while c.dim() < a.dim():                    # a vector condition selects rows (TF semantics)
        c = c.unsqueeze(-1)
    return Tensor(torch.where(c, a, b))


def gather(params, indices, name=None, axis=0):
    return Tensor(torch.index_select(_raw(params), axis, _raw(indices).reshape(-1).long()).reshape(
        list(_raw(indices).shape) + list(_raw(params).shape[1:])))


def reverse_sequence(input, seq_lengths, seq_axis=None, batch_axis=None, name=None, seq_dim=None, batch_dim=None):
    """Reverses the first seq_lengths[b] entries along seq_axis of every batch row, the rest stays in place."""
    seq_axis = seq_axis if seq_axis is not None else seq_dim
    batch_axis = batch_axis if batch_axis is not None else (batch_dim or 0)
    t = _raw(input)
    L = _raw(seq_lengths).long()
    x = t.transpose(0, seq_axis) if seq_axis != 0 else t          # [S, ...]
    b_ax = batch_axis if seq_axis == 0 else (0 if batch_axis == seq_axis else batch_axis)
    if seq_axis != 0 and batch_axis == 0:
        b_ax = seq_axis
    x = x.movedim(b_ax, 1)                                         # [S, B, ...]
    S, B = x.shape[0], x.shape[1]
    pos = torch.arange(S).unsqueeze(1)
    src = torch.where(pos < L.unsqueeze(0), L.unsqueeze(0) - 1 - pos, pos)
    x = x[src, torch.arange(B).unsqueeze(0)]
    x = x.movedim(1, b_ax)
    return Tensor(x.transpose(0, seq_axis) if seq_axis != 0 else x)


def sequence_mask(lengths, maxlen=None, dtype=None, name=None):
    L = _raw(lengths).long()
    m = int(_raw(maxlen)) if maxlen is not None else int(L.max())
    return Tensor((torch.arange(m).unsqueeze(0) < L.unsqueeze(-1)).to(_tdtype(dtype, torch.bool)))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    x, y = _raw(a), _raw(b)
    if transpose_a:
        x = x.transpose(-1, -2)
    if transpose_b:
        y = y.transpose(-1, -2)
    return Tensor(torch.matmul(x, y))


def tensordot(a, b, axes, name=None):
    return Tensor(torch.tensordot(_raw(a), _raw(b), dims=axes))


def add(a, b, name=None): return convert_to_tensor(a) + b
def subtract(a, b, name=None): return convert_to_tensor(a) - b
def multiply(a, b, name=None): return convert_to_tensor(a) * b
def divide(a, b, name=None): return convert_to_tensor(a) / b
def add_n(xs, name=None):
    out = xs[0]
    for x in xs[1:]:
        out = out + x
    return convert_to_tensor(out)


def sigmoid(x, name=None): return Tensor(torch.sigmoid(_raw(x)))
def tanh(x, name=None): return Tensor(torch.tanh(_raw(x)))
def exp(x, name=None): return Tensor(torch.exp(_raw(x)))
def log(x, name=None): return Tensor(torch.log(_raw(x)))
def sqrt(x, name=None): return Tensor(torch.sqrt(_raw(x)))
def square(x, name=None): return Tensor(_raw(x) ** 2)
def abs(x, name=None): return Tensor(torch.abs(_raw(x)))          # noqa: A001
def maximum(a, b, name=None): return Tensor(torch.maximum(_raw(a).to(_F), _raw(b).to(_F)))
def minimum(a, b, name=None): return Tensor(torch.minimum(_raw(a).to(_F), _raw(b).to(_F)))
def equal(a, b, name=None): return Tensor(torch.eq(_raw(a), _raw(b)))
def not_equal(a, b, name=None): return Tensor(torch.ne(_raw(a), _raw(b)))
def greater_equal(a, b, name=None): return Tensor(torch.ge(_raw(a), _raw(b)))
def greater(a, b, name=None): return Tensor(torch.gt(_raw(a), _raw(b)))
def less(a, b, name=None): return Tensor(torch.lt(_raw(a), _raw(b)))
def less_equal(a, b, name=None): return Tensor(torch.le(_raw(a), _raw(b)))
def logical_or(a, b, name=None): return Tensor(torch.logical_or(_raw(a), _raw(b)))
def logical_and(a, b, name=None): return Tensor(torch.logical_and(_raw(a), _raw(b)))
def logical_not(a, name=None): return Tensor(torch.logical_not(_raw(a)))


def _reduce(fn, x, axis, keep_dims):
    t = _raw(x)
    if axis is None:
        return Tensor(fn(t))
    out = t
    for a in sorted([d % t.dim() for d in _ints(axis)], reverse=True):
        out = fn(out, dim=a, keepdim=keep_dims)
        if isinstance(out, tuple):
            out = out[0]
    return Tensor(out)


def reduce_sum(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    return _reduce(torch.sum, x, axis if axis is not None else reduction_indices, keep_dims or _py_bool(keepdims))


def reduce_mean(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    return _reduce(torch.mean, x, axis if axis is not None else reduction_indices, keep_dims or _py_bool(keepdims))


def reduce_max(x, axis=None, keep_dims=False, name=None, reduction_indices=None):
    return _reduce(lambda t, **k: torch.max(t, **k) if k else torch.max(t), x,
                   axis if axis is not None else reduction_indices, keep_dims)


def reduce_min(x, axis=None, keep_dims=False, name=None, reduction_indices=None):
    return _reduce(lambda t, **k: torch.min(t, **k) if k else torch.min(t), x,
                   axis if axis is not None else reduction_indices, keep_dims)


def reduce_all(x, axis=None, keep_dims=False, name=None):
    assert axis is None
    return Tensor(torch.all(_raw(x)))


def reduce_any(x, axis=None, keep_dims=False, name=None):
    assert axis is None
    return Tensor(torch.any(_raw(x)))


def argmax(x, axis=None, name=None, dimension=None, output_type=None):
    axis = axis if axis is not None else (dimension if dimension is not None else 0)
    t = _raw(x)
    # first maximum on ties, as TF / numpy (torch.argmax does not promise it)
    return Tensor(torch.from_numpy(np.argmax(t.detach().numpy(), axis=axis)))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    """min(max(t, lo), hi); gradient passes where the value was not clamped (tf.clip_by_value)."""
    return Tensor(torch.clamp(_raw(t), float(clip_value_min), float(clip_value_max)))


def clip_by_norm(t, clip_norm, axes=None, name=None):
    """t * clip_norm / max(||t||_2, clip_norm)  (tf.clip_by_norm, clip_ops.py)."""
    x = _raw(t)
    l2 = torch.sqrt(torch.sum(x * x))
    return Tensor(x * clip_norm / torch.maximum(l2, torch.tensor(float(clip_norm), dtype=_F)))


def stop_gradient(x, name=None):
    return Tensor(_raw(x).detach())


def one_hot(indices, depth, on_value=1.0, off_value=0.0, axis=-1, dtype=None, name=None):
    return Tensor(torch.nn.functional.one_hot(_raw(indices).long(), int(depth)).to(_F) * (on_value - off_value)
                  + off_value)


def cond(pred, true_fn=None, false_fn=None, name=None, fn1=None, fn2=None, strict=False):
    return (true_fn or fn1)() if _py_bool(_raw(pred)) else (false_fn or fn2)()


def while_loop(cond, body, loop_vars, shape_invariants=None, parallel_iterations=10, back_prop=True,
               swap_memory=False, name=None):
    """Eager: the body runs once per iteration (a TF1 graph traces it once; see the module docstring)."""
    vars_ = list(loop_vars)
    while _py_bool(_raw(cond(*vars_))):
        vars_ = list(body(*vars_))
    return vars_


class TensorArray(object):
    def __init__(self, dtype, size=None, dynamic_size=None, clear_after_read=None, tensor_array_name=None,
                 element_shape=None, infer_shape=True, name=None, **kw):
        self.dtype = dtype
        self._items = {}

    def write(self, index, value, name=None):
        self._items[int(_raw(index))] = convert_to_tensor(value)
        return self

    def read(self, index, name=None):
        return self._items[int(_raw(index))]

    def stack(self, name=None):
        return Tensor(torch.stack([self._items[i]._t for i in sorted(self._items)], dim=0))

    def unstack(self, value, name=None):
        for i, t in enumerate(torch.unbind(_raw(value), dim=0)):
            self._items[i] = Tensor(t)
        return self

    def size(self):
        return Tensor(torch.tensor(len(self._items)))


class SparseTensor(object):
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = indices, values, dense_shape


def placeholder(dtype, shape=None, name=None):
    return None


def constant_value(tensor, partial=False):
    return None if tensor is None else np.asarray(_raw(tensor).detach().numpy())


# ----------------------------------------------------------------------------------------------------------------
# nest (tensorflow.python.util.nest): tuples / lists / namedtuples / dicts are structure, everything else a leaf
def _is_seq(x):
    return isinstance(x, (tuple, list, dict)) and not isinstance(x, (str, bytes))


def _like(template, items):
    if isinstance(template, dict):
        return type(template)(zip(sorted(template), items))
    if isinstance(template, tuple) and hasattr(template, '_fields'):
        return type(template)(*items)
    return type(template)(items)


def _children(x):
    return [x[k] for k in sorted(x)] if isinstance(x, dict) else list(x)


def nest_flatten(x):
    if not _is_seq(x):
        return [x]
    out = []
    for c in _children(x):
        out.extend(nest_flatten(c))
    return out


def nest_assert_same_structure(a, b, check_types=True):
    if _is_seq(a) != _is_seq(b):
        raise ValueError('structures differ: %r vs %r' % (a, b))
    if _is_seq(a):
        ca, cb = _children(a), _children(b)
        if len(ca) != len(cb):
            raise ValueError('structures differ: %r vs %r' % (a, b))
        for x, y in zip(ca, cb):
            nest_assert_same_structure(x, y, check_types)


def nest_map_structure(fn, *structs, **kw):
    first = structs[0]
    if not _is_seq(first):
        return fn(*structs)
    kids = [_children(s) for s in structs]
    return _like(first, [nest_map_structure(fn, *ks) for ks in zip(*kids)])


def nest_pack_sequence_as(structure, flat):
    flat = list(flat)

    def build(s):
        if not _is_seq(s):
            return flat.pop(0)
        return _like(s, [build(c) for c in _children(s)])
    return build(structure)


# ----------------------------------------------------------------------------------------------------------------
# tf.nn
def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2        # the extra element goes AFTER


def nn_softmax(logits, dim=-1, name=None, axis=None):
    return Tensor(torch.softmax(_raw(logits), dim=axis if axis is not None else dim))


def nn_log_softmax(logits, dim=-1, name=None, axis=None):
    return Tensor(torch.log_softmax(_raw(logits), dim=axis if axis is not None else dim))


def nn_relu(x, name=None):
    return Tensor(torch.relu(_raw(x)))


def nn_bias_add(value, bias, data_format=None, name=None):
    return Tensor(_raw(value) + _raw(bias))


def nn_l2_loss(t, name=None):
    return Tensor(torch.sum(_raw(t) ** 2) / 2)


def nn_dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    kp = float(_raw(keep_prob))
    if kp != 1.0:
        raise NotImplementedError('the fixtures are generated without dropout (keep_prob == 1)')
    return convert_to_tensor(x)


def nn_embedding_lookup(params, ids, partition_strategy='mod', name=None, validate_indices=True, max_norm=None):
    return Tensor(_raw(params)[_raw(ids).long()])


def nn_conv1d(value, filters, stride, padding, use_cudnn_on_gpu=None, data_format=None, name=None):
    """value [B, W, Cin], filters [K, Cin, Cout]: cross-correlation (no kernel flip), SAME = ceil(W / stride)."""
    x, f = _raw(value), _raw(filters)
    if padding == 'SAME':
        lo, hi = _same_pad(x.shape[1], f.shape[0], stride)
        x = torch.nn.functional.pad(x, (0, 0, lo, hi))
    y = torch.nn.functional.conv1d(x.transpose(1, 2), f.permute(2, 1, 0), stride=stride)
    return Tensor(y.transpose(1, 2))


def nn_conv2d(input, filter, strides, padding, use_cudnn_on_gpu=None, data_format=None, name=None):
    """input NHWC, filter HWIO, cross-correlation."""
    x, f = _raw(input), _raw(filter)
    sh, sw = strides[1], strides[2]
    if padding == 'SAME':
        t, b = _same_pad(x.shape[1], f.shape[0], sh)
        l, r = _same_pad(x.shape[2], f.shape[1], sw)
        x = torch.nn.functional.pad(x, (0, 0, l, r, t, b))
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), f.permute(3, 2, 0, 1), stride=(sh, sw))
    return Tensor(y.permute(0, 2, 3, 1))


def _pool(value, ksize, strides, padding, fn, pad_value):
    x = _raw(value)
    kh, kw, sh, sw = ksize[1], ksize[2], strides[1], strides[2]
    if padding == 'SAME':
        t, b = _same_pad(x.shape[1], kh, sh)
        l, r = _same_pad(x.shape[2], kw, sw)
        x = torch.nn.functional.pad(x, (0, 0, l, r, t, b), value=pad_value)
    return Tensor(fn(x.permute(0, 3, 1, 2), (kh, kw), (sh, sw)).permute(0, 2, 3, 1))


def nn_max_pool(value, ksize, strides, padding, data_format='NHWC', name=None):
    return _pool(value, ksize, strides, padding, torch.nn.functional.max_pool2d, float('-inf'))


def nn_ctc_loss(labels, inputs, sequence_length, preprocess_collapse_repeated=False, ctc_merge_repeated=True,
                ignore_longer_outputs_than_inputs=False, time_major=True):
    """torch.nn.functional.ctc_loss, float64, blank = num_classes - 1 (tf.nn.ctc_loss's convention)."""
    assert not preprocess_collapse_repeated and ctc_merge_repeated
    lg = _raw(inputs)
    if not time_major:
        lg = lg.transpose(0, 1)
    T, B, C = lg.shape
    idx = np.asarray(_raw(labels.indices).numpy()).reshape(-1, 2)
    vals = np.asarray(_raw(labels.values).numpy()).reshape(-1)
    rows = [[] for _ in _py_range(B)]
    for (b, _), v in zip(idx, vals):
        rows[int(b)].append(int(v))
    tl = torch.tensor([len(r) for r in rows], dtype=torch.long)
    tg = torch.tensor([v for r in rows for v in r], dtype=torch.long)
    il = _raw(sequence_length).long()
    losses = torch.nn.functional.ctc_loss(torch.log_softmax(lg, dim=2), tg, il, tl, blank=C - 1, reduction='none',
                                          zero_infinity=True)
    if not ignore_longer_outputs_than_inputs:
        for b in _py_range(B):
            need = len(rows[b]) + sum(1 for i in _py_range(1, len(rows[b])) if rows[b][i] == rows[b][i - 1])
            if need > int(il[b]):
                raise ValueError('Not enough time for target transition sequence')
    return Tensor(losses)


def _dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None, parallel_iterations=None,
                 swap_memory=False, time_major=False, scope=None):
    """tf.nn.dynamic_rnn: past sequence_length[b] the emitted output is zero and the state is copied through
    (rnn.py _rnn_step); final state = the state at frame sequence_length[b] - 1."""
    x = _raw(inputs)
    if not time_major:
        x = x.transpose(0, 1)
    T, B = x.shape[0], x.shape[1]
    with variable_scope(scope or 'rnn'):
        state = initial_state if initial_state is not None else cell.zero_state(B, dtype)
        L = _raw(sequence_length).long() if sequence_length is not None else None
        outs = []
        for t in _py_range(T):
            out, new_state = cell(Tensor(x[t]), state)
            if L is not None:
                done = Tensor(t >= L)
                out = where(done, zeros_like(out), out)
                new_state = nest_map_structure(lambda n, o: where(done, o, n), new_state, state)
            state = new_state
            outs.append(out._t)
    y = torch.stack(outs, dim=0)
    if not time_major:
        y = y.transpose(0, 1)
    return Tensor(y), state


def _bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, sequence_length=None, initial_state_fw=None,
                               initial_state_bw=None, dtype=None, parallel_iterations=None, swap_memory=False,
                               time_major=False, scope=None):
    """tf.nn.bidirectional_dynamic_rnn: scopes fw / bw; the backward cell runs over reverse_sequence(inputs) and its
    outputs are reversed back."""
    t_ax, b_ax = (0, 1) if time_major else (1, 0)
    with variable_scope(scope or 'bidirectional_rnn'):
        with variable_scope('fw') as fw_scope:
            out_fw, st_fw = _dynamic_rnn(cell_fw, inputs, sequence_length, initial_state_fw, dtype,
                                         time_major=time_major, scope=fw_scope)
        rev = reverse_sequence(inputs, sequence_length, seq_axis=t_ax, batch_axis=b_ax)
        with variable_scope('bw') as bw_scope:
            tmp, st_bw = _dynamic_rnn(cell_bw, rev, sequence_length, initial_state_bw, dtype,
                                      time_major=time_major, scope=bw_scope)
        out_bw = reverse_sequence(tmp, sequence_length, seq_axis=t_ax, batch_axis=b_ax)
    return (out_fw, out_bw), (st_fw, st_bw)


# ----------------------------------------------------------------------------------------------------------------
# tf.contrib.layers / rnn / seq2seq
def fully_connected(inputs, num_outputs, activation_fn=nn_relu, normalizer_fn=None, normalizer_params=None,
                    weights_initializer=None, weights_regularizer=None, biases_initializer=zeros_initializer(),
                    biases_regularizer=None, reuse=None, variables_collections=None, outputs_collections=None,
                    trainable=True, scope=None):
    """tf.contrib.layers.fully_connected: variables `weights` [in, out] and `biases` [out] under `scope` (default
    name 'fully_connected'), applied to the last axis; activation_fn defaults to relu."""
    x = _raw(inputs)
    with variable_scope(scope, 'fully_connected', reuse=reuse):
        w = get_variable('weights', [x.shape[-1], int(num_outputs)], initializer=weights_initializer)
        y = torch.matmul(x, w._t)
        if biases_initializer is not None:
            b = get_variable('biases', [int(num_outputs)], initializer=biases_initializer)
            y = y + b._t
    out = Tensor(y)
    return activation_fn(out) if activation_fn is not None else out


LSTMStateTuple = collections.namedtuple('LSTMStateTuple', ('c', 'h'))


def _linear(args, output_size, bias, bias_start=0.0, bias_initializer=None, kernel_initializer=None):
    """rnn_cell_impl._linear: concat(args, 1) . kernel [+ bias]; variables `kernel`, `bias` (TF >= 1.2 names)."""
    if not isinstance(args, (list, tuple)):
        args = [args]
    x = torch.cat([_raw(a) for a in args], dim=1)
    w = get_variable('kernel', [x.shape[1], int(output_size)], initializer=kernel_initializer)
    y = x @ w._t
    if bias:
        b = get_variable('bias', [int(output_size)],
                         initializer=bias_initializer or constant_initializer(bias_start))
        y = y + b._t
    return Tensor(y)


class RNNCell(object):
    def zero_state(self, batch_size, dtype):
        bs = int(_raw(batch_size))
        return nest_map_structure(lambda s: Tensor(torch.zeros(bs, int(s), dtype=_F)), self.state_size)


class LSTMBlockCell(RNNCell):
    """tf.contrib.rnn.LSTMBlockCell (TF 1.3 lstm_ops.py): xh = [x, h_prev]; icfo = xh.W + b with
    column blocks i, ci, f, o; i = sig(i + wci*cs_prev); ci = tanh(ci); f = sig(f + forget_bias + wcf*cs_prev);
    cs = ci*i + cs_prev*f; clip to +-cell_clip; o = sig(o + wco*cs); h = tanh(cs)*o.  The gradient kernel
    (LSTMBlockCellGrad) has no clip attribute: the clamp is transparent to the gradient.  The generator checks this
    forward against the reference's own Python statement of the same cell (models/recurrent/layers/lstm.py:142-170)."""

    def __init__(self, num_units, forget_bias=1.0, clip_cell=None, use_peephole=False, cell_clip=None, reuse=None):
        self._num_units, self._forget_bias, self._use_peephole = num_units, forget_bias, use_peephole
        self._cell_clip = clip_cell if clip_cell is not None else cell_clip

    @property
    def state_size(self):
        return LSTMStateTuple(self._num_units, self._num_units)

    @property
    def output_size(self):
        return self._num_units

    def __call__(self, x, states_prev, scope=None):
        H = self._num_units
        cs_prev, h_prev = states_prev
        xr = _raw(x)
        with variable_scope(scope or 'lstm_cell'):
            w = get_variable('kernel', [xr.shape[1] + H, 4 * H])
            b = get_variable('bias', [4 * H], initializer=constant_initializer(0.0))
            if self._use_peephole:
                wci = get_variable('w_i_diag', [H])._t
                wcf = get_variable('w_f_diag', [H])._t
                wco = get_variable('w_o_diag', [H])._t
            else:
                wci = wcf = wco = torch.zeros(H, dtype=_F)
        icfo = torch.cat([xr, _raw(h_prev)], dim=1) @ w._t + b._t
        i, ci, f, o = torch.split(icfo, H, dim=1)
        c0 = _raw(cs_prev)
        i = torch.sigmoid(i + wci * c0)
        ci = torch.tanh(ci)
        f = torch.sigmoid(f + self._forget_bias + wcf * c0)
        cs = ci * i + c0 * f
        if self._cell_clip is not None and self._cell_clip > 0:
            cs = cs + (torch.clamp(cs, -self._cell_clip, self._cell_clip) - cs).detach()
        o = torch.sigmoid(o + wco * cs)
        h = torch.tanh(cs) * o
        return Tensor(h), LSTMStateTuple(Tensor(cs), Tensor(h))


class BasicLSTMCell(RNNCell):
    """tf.contrib.rnn.BasicLSTMCell: i, j, f, o = split([x, h].kernel + bias); c = c*sig(f + fb) + sig(i)*tanh(j)."""

    def __init__(self, num_units, forget_bias=1.0, state_is_tuple=True, activation=None, reuse=None):
        self._num_units, self._forget_bias = num_units, forget_bias

    @property
    def state_size(self):
        return LSTMStateTuple(self._num_units, self._num_units)

    @property
    def output_size(self):
        return self._num_units

    def __call__(self, x, state, scope=None):
        c, h = state
        with variable_scope(scope or 'basic_lstm_cell'):
            i, j, f, o = split(_linear([x, h], 4 * self._num_units, True), 4, 1)
        new_c = c * sigmoid(f + self._forget_bias) + sigmoid(i) * tanh(j)
        new_h = tanh(new_c) * sigmoid(o)
        return new_h, LSTMStateTuple(new_c, new_h)


class GRUCell(RNNCell):
    """tf.contrib.rnn.GRUCell (TF 1.3): [r, u] = sig([x, h].W_g + b_g) with b_g starting at 1; c = tanh([x, r*h].W_c +
    b_c); h' = u*h + (1 - u)*c."""

    def __init__(self, num_units, activation=None, reuse=None, kernel_initializer=None, bias_initializer=None):
        self._num_units = num_units

    @property
    def state_size(self):
        return self._num_units

    @property
    def output_size(self):
        return self._num_units

    def __call__(self, x, state, scope=None):
        with variable_scope(scope or 'gru_cell'):
            with variable_scope('gates'):
                ru = sigmoid(_linear([x, state], 2 * self._num_units, True, 1.0))
                r, u = split(ru, 2, 1)
            with variable_scope('candidate'):
                c = tanh(_linear([x, r * state], self._num_units, True))
        new_h = u * state + (1 - u) * c
        return new_h, new_h


class DropoutWrapper(RNNCell):
    def __init__(self, cell, input_keep_prob=1.0, output_keep_prob=1.0, state_keep_prob=1.0, **kw):
        for kp in (input_keep_prob, output_keep_prob, state_keep_prob):
            if float(_raw(kp)) != 1.0:
                raise NotImplementedError('the fixtures are generated without dropout (keep_prob == 1)')
        self._cell = cell

    @property
    def state_size(self):
        return self._cell.state_size

    @property
    def output_size(self):
        return self._cell.output_size

    def zero_state(self, batch_size, dtype):
        return self._cell.zero_state(batch_size, dtype)

    def __call__(self, x, state, scope=None):
        return self._cell(x, state, scope)


class MultiRNNCell(RNNCell):
    """tf.contrib.rnn.MultiRNNCell: cell i runs under multi_rnn_cell/cell_{i} on the output of cell i - 1."""

    def __init__(self, cells, state_is_tuple=True):
        self._cells = list(cells)

    @property
    def state_size(self):
        return tuple(c.state_size for c in self._cells)

    @property
    def output_size(self):
        return self._cells[-1].output_size

    def zero_state(self, batch_size, dtype):
        return tuple(c.zero_state(batch_size, dtype) for c in self._cells)

    def __call__(self, x, state, scope=None):
        new_states = []
        with variable_scope(scope or 'multi_rnn_cell'):
            for i, cell in enumerate(self._cells):
                with variable_scope('cell_%d' % i):
                    x, ns = cell(x, state[i])
                    new_states.append(ns)
        return x, tuple(new_states)


class Decoder(object):
    """tf.contrib.seq2seq.Decoder (abstract)."""

    def __init__(self, *a, **k):
        pass


class Helper(object):
    pass


class CustomHelper(Helper):
    def __init__(self, initialize_fn, sample_fn, next_inputs_fn):
        self._initialize_fn, self._sample_fn, self._next_inputs_fn = initialize_fn, sample_fn, next_inputs_fn
        self._batch_size = None

    @property
    def batch_size(self):
        return self._batch_size

    def initialize(self, name=None):
        finished, next_inputs = self._initialize_fn()
        self._batch_size = size(finished)
        return finished, next_inputs

    def sample(self, time, outputs, state, name=None):
        return self._sample_fn(time=time, outputs=outputs, state=state)

    def next_inputs(self, time, outputs, state, sample_ids, name=None):
        return self._next_inputs_fn(time=time, outputs=outputs, state=state, sample_ids=sample_ids)


class TrainingHelper(Helper):
    """tf.contrib.seq2seq.TrainingHelper (TF 1.3 helper.py): finished = (time + 1 >= sequence_length); next inputs =
    inputs[time + 1], or zeros once EVERY row has finished; sample = argmax of the outputs."""

    def __init__(self, inputs, sequence_length, time_major=False, name=None):
        x = _raw(inputs)
        self._inputs = x if time_major else x.transpose(0, 1)          # [T, B, E]
        self._sequence_length = _raw(sequence_length).long()
        self._zero = torch.zeros_like(self._inputs[0])
        self._batch_size = size(sequence_length)

    @property
    def batch_size(self):
        return self._batch_size

    def initialize(self, name=None):
        finished = self._sequence_length == 0
        nxt = self._zero if _py_bool(finished.all()) else self._inputs[0]
        return Tensor(finished), Tensor(nxt)

    def sample(self, time, outputs, name=None, **unused):
        return argmax(outputs, axis=-1)

    def next_inputs(self, time, outputs, state, name=None, **unused):
        nt = int(_raw(time)) + 1
        finished = nt >= self._sequence_length
        nxt = self._zero if _py_bool(finished.all()) else self._inputs[nt]
        return Tensor(finished), Tensor(nxt), state


class GreedyEmbeddingHelper(Helper):
    """tf.contrib.seq2seq.GreedyEmbeddingHelper: sample = argmax; finished = (sample == end_token); next inputs =
    embedding[sample], or the start inputs once every row has finished."""

    def __init__(self, embedding, start_tokens, end_token):
        self._emb = (lambda ids: nn_embedding_lookup(embedding, ids)) if not callable(embedding) else embedding
        self._start_tokens = convert_to_tensor(start_tokens)
        self._end_token = int(_raw(end_token))
        self._start_inputs = self._emb(self._start_tokens)
        self._batch_size = size(start_tokens)

    @property
    def batch_size(self):
        return self._batch_size

    def initialize(self, name=None):
        return Tensor(torch.zeros(int(self._batch_size), dtype=torch.bool)), self._start_inputs

    def sample(self, time, outputs, state, name=None):
        return argmax(outputs, axis=-1)

    def next_inputs(self, time, outputs, state, sample_ids, name=None):
        finished = _raw(sample_ids) == self._end_token
        nxt = self._start_inputs if _py_bool(finished.all()) else self._emb(sample_ids)
        return Tensor(finished), nxt, state


def sequence_loss(logits, targets, weights, average_across_timesteps=True, average_across_batch=True,
                  softmax_loss_function=None, name=None):
    """tf.contrib.seq2seq.sequence_loss: sum(w * sparse softmax cross-entropy) / (sum(w) + 1e-12)."""
    assert average_across_timesteps and average_across_batch and softmax_loss_function is None
    lg, tg, w = _raw(logits), _raw(targets).long(), _raw(weights)
    C = lg.shape[-1]
    x = torch.nn.functional.cross_entropy(lg.reshape(-1, C), tg.reshape(-1), reduction='none') * w.reshape(-1)
    return Tensor(x.sum() / (w.sum() + 1e-12))


class ModeKeys(object):
    TRAIN, EVAL, INFER = 'train', 'eval', 'infer'


# ----------------------------------------------------------------------------------------------------------------
# tf.train: optimizers with TF1's update rules and defaults (compute_gradients = autograd of the eager graph)
class _Optimizer(object):
    def __init__(self, learning_rate, **kw):
        self._lr = learning_rate
        self._slots = {}

    def compute_gradients(self, loss, var_list=None, **kw):
        vs = var_list or trainable_variables()
        gs = torch.autograd.grad(_raw(loss), [v._t for v in vs], allow_unused=True, retain_graph=True)
        return [(None if g is None else Tensor(g), v) for g, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        lr = float(_raw(self._lr))
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is not None:
                    self._apply(v, _raw(g), lr, self._slots.setdefault(v.name, {}))
        if global_step is not None:
            global_step._t = global_step._t + 1
        return None

    def minimize(self, loss, global_step=None, var_list=None, **kw):
        return self.apply_gradients(self.compute_gradients(loss, var_list), global_step)


class GradientDescentOptimizer(_Optimizer):
    def _apply(self, v, g, lr, s):
        v._t -= lr * g


class MomentumOptimizer(_Optimizer):
    def __init__(self, learning_rate, momentum, use_locking=False, name='Momentum', use_nesterov=False):
        _Optimizer.__init__(self, learning_rate)
        self._m, self._nesterov = momentum, use_nesterov

    def _apply(self, v, g, lr, s):
        acc = s.setdefault('m', torch.zeros_like(g))
        acc.mul_(self._m).add_(g)
        v._t -= lr * (g + self._m * acc) if self._nesterov else lr * acc


class AdagradOptimizer(_Optimizer):
    def __init__(self, learning_rate, initial_accumulator_value=0.1, **kw):
        _Optimizer.__init__(self, learning_rate)
        self._init = initial_accumulator_value

    def _apply(self, v, g, lr, s):
        acc = s.setdefault('a', torch.full_like(g, self._init))
        acc.add_(g * g)
        v._t -= lr * g / torch.sqrt(acc)


class AdadeltaOptimizer(_Optimizer):
    def __init__(self, learning_rate=0.001, rho=0.95, epsilon=1e-8, **kw):
        _Optimizer.__init__(self, learning_rate)
        self._rho, self._eps = rho, epsilon

    def _apply(self, v, g, lr, s):
        acc = s.setdefault('a', torch.zeros_like(g))
        upd = s.setdefault('u', torch.zeros_like(g))
        acc.mul_(self._rho).add_((1 - self._rho) * g * g)
        u = torch.sqrt(upd + self._eps) / torch.sqrt(acc + self._eps) * g
        upd.mul_(self._rho).add_((1 - self._rho) * u * u)
        v._t -= lr * u


class RMSPropOptimizer(_Optimizer):
    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, **kw):
        _Optimizer.__init__(self, learning_rate)
        self._decay, self._mom, self._eps = decay, momentum, epsilon

    def _apply(self, v, g, lr, s):
        rms = s.setdefault('r', torch.ones_like(g))
        mom = s.setdefault('m', torch.zeros_like(g))
        rms.mul_(self._decay).add_((1 - self._decay) * g * g)
        mom.mul_(self._mom).add_(lr * g / torch.sqrt(rms + self._eps))
        v._t -= mom


class AdamOptimizer(_Optimizer):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw):
        _Optimizer.__init__(self, learning_rate)
        self._b1, self._b2, self._eps = beta1, beta2, epsilon

    def _apply(self, v, g, lr, s):
        m = s.setdefault('m', torch.zeros_like(g))
        vv = s.setdefault('v', torch.zeros_like(g))
        s['t'] = s.get('t', 0) + 1
        m.mul_(self._b1).add_((1 - self._b1) * g)
        vv.mul_(self._b2).add_((1 - self._b2) * g * g)
        lr_t = lr * math.sqrt(1 - self._b2 ** s['t']) / (1 - self._b1 ** s['t'])
        v._t -= lr_t * m / (torch.sqrt(vv) + self._eps)


# ----------------------------------------------------------------------------------------------------------------
# module tree


class _Missing(object):
    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise NotImplementedError('tensorflow shim: %s is not restated' % self._name)

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Missing(self._name + '.' + n)


class _Module(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Missing(self.__name__ + '.' + n)


def _mod(name, **attrs):
    m = _Module(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Any `tensorflow.x.y` that is not built below imports as an empty stub whose attributes raise when CALLED (the
    reference's module-level imports of TF internals it never reaches must not fail)."""

    def find_spec(self, fullname, path, target=None):
        if fullname.startswith('tensorflow.') and fullname not in sys.modules:
            parent, _, leaf = fullname.rpartition('.')
            if parent in sys.modules and leaf in sys.modules[parent].__dict__:
                return None               # an attribute, not a module (pydoc.locate('tensorflow.identity'))
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Module(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_G = _Graph()
_me = sys.modules[__name__]
sys.meta_path.insert(0, _Finder())

_everything = {k: v for k, v in list(globals().items()) if not k.startswith('_')}

nn = _mod('tensorflow.nn', softmax=nn_softmax, log_softmax=nn_log_softmax, relu=nn_relu, tanh=tanh, sigmoid=sigmoid,
          bias_add=nn_bias_add, l2_loss=nn_l2_loss, dropout=nn_dropout, embedding_lookup=nn_embedding_lookup,
          conv1d=nn_conv1d, conv2d=nn_conv2d, max_pool=nn_max_pool, ctc_loss=nn_ctc_loss,
          dynamic_rnn=_dynamic_rnn, bidirectional_dynamic_rnn=_bidirectional_dynamic_rnn)
train = _mod('tensorflow.train', GradientDescentOptimizer=GradientDescentOptimizer,
             MomentumOptimizer=MomentumOptimizer, AdagradOptimizer=AdagradOptimizer,
             AdadeltaOptimizer=AdadeltaOptimizer, RMSPropOptimizer=RMSPropOptimizer, AdamOptimizer=AdamOptimizer)
summary = _mod('tensorflow.summary', scalar=lambda *a, **k: None, histogram=lambda *a, **k: None,
               merge=lambda *a, **k: None, merge_all=lambda *a, **k: None)
contrib = _mod('tensorflow.contrib')
_mod('tensorflow.contrib.layers', fully_connected=fully_connected, xavier_initializer=glorot_uniform_initializer)
_mod('tensorflow.contrib.rnn', RNNCell=RNNCell, LSTMStateTuple=LSTMStateTuple, LSTMBlockCell=LSTMBlockCell,
     BasicLSTMCell=BasicLSTMCell, GRUCell=GRUCell, DropoutWrapper=DropoutWrapper, MultiRNNCell=MultiRNNCell,
     _linear=_linear)
_mod('tensorflow.contrib.seq2seq', Decoder=Decoder, Helper=Helper, CustomHelper=CustomHelper,
     TrainingHelper=TrainingHelper, GreedyEmbeddingHelper=GreedyEmbeddingHelper, sequence_loss=sequence_loss)
_mod('tensorflow.contrib.learn', ModeKeys=ModeKeys)
_mod('tensorflow.python')
_mod('tensorflow.python.util')
_mod('tensorflow.python.util.nest', flatten=nest_flatten, map_structure=nest_map_structure,
     assert_same_structure=nest_assert_same_structure, pack_sequence_as=nest_pack_sequence_as, is_sequence=_is_seq)
_mod('tensorflow.python.platform')
_mod('tensorflow.python.platform.tf_logging', warn=lambda *a, **k: None, info=lambda *a, **k: None,
     warning=lambda *a, **k: None)
_mod('tensorflow.python.framework')
_mod('tensorflow.python.framework.constant_op', constant=constant)
_mod('tensorflow.python.framework.dtypes', float32=float32, float64=float64, int32=int32, int64=int64, bool=bool,
     DType=DType)
_mod('tensorflow.python.framework.ops', Tensor=Tensor, convert_to_tensor=convert_to_tensor, name_scope=name_scope)
_mod('tensorflow.python.framework.tensor_shape', TensorShape=TensorShape, Dimension=Dimension)
_mod('tensorflow.python.framework.tensor_util', constant_value=constant_value)
_mod('tensorflow.python.ops')
_mod('tensorflow.python.ops.array_ops', **_everything)
_mod('tensorflow.python.ops.math_ops', **_everything)
_mod('tensorflow.python.ops.control_flow_ops', while_loop=while_loop, cond=cond)
_mod('tensorflow.python.ops.tensor_array_ops', TensorArray=TensorArray)
_mod('tensorflow.python.ops.variable_scope', variable_scope=variable_scope, get_variable=get_variable,
     get_variable_scope=get_variable_scope, VariableScope=VariableScope)
