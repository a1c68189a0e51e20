"""Thin torch-tensor front end of the C ABI (include/asr_hip.h).

torch owns device memory and the stream; every function below hands raw device
pointers + shapes to libasr_hip.so.  No arithmetic happens in torch here and
there is no CPU fallback: non-CUDA tensors raise.
"""
import ctypes as C
import os
import time as _time

import numpy as np
import torch

from . import _lib
from ._lib import ASR_BF16, ASR_F32

TORCH_DTYPE = {ASR_F32: torch.float32, ASR_BF16: torch.bfloat16}


def dtype_id(dtype):
    if dtype in (ASR_F32, 'f32', 'fp32', 'float32', torch.float32):
        return ASR_F32
    if dtype in (ASR_BF16, 'bf16', 'bfloat16', torch.bfloat16):
        return ASR_BF16
    raise ValueError('dtype must be f32 or bf16, got %r' % (dtype,))


_lane = 0            # 0: main stream / handle, 1: side stream / second handle (side_lane)
_side = {}           # device index -> dict(stream, keep)


def _h(t):
    if not t.is_cuda:
        raise RuntimeError('HIP path needs a CUDA(ROCm) tensor; there is no CPU fallback')
    return _lib.handle(t.device.index or 0, _lane)


def debug_placement(nblocks, stream=None, spin_cycles=200000, device=0):
    """[nblocks, 2] int32 (XCC id, HW_ID) of a probe grid launched on `stream` (default: current)."""
    h = _lib.handle(device, 0)
    out = torch.zeros((nblocks, 2), dtype=torch.int32, device='cuda:%d' % device)
    sp = C.c_void_p(stream.cuda_stream) if stream is not None else _s()
    h.check(h.lib.asr_debug_placement(h.h, _p(out), nblocks, spin_cycles, sp), 'asr_debug_placement')
    return out


class side_lane(object):
    """Issue the enclosed launches on a side HIP stream (with its own handle, i.e. its own
    split-K scratch) ordered after everything already on the main stream.  Used for work
    nothing downstream waits on until later (weight images of the layers ahead, weight-gradient
    GEMMs), so it overlaps with the recurrence kernels, which only occupy a handful of CUs.
    `lane` 1, 2, ...: independent side streams (the two directions of a layer's weight gradients
    run side by side on lanes 1 and 2).
    `keep`: tensors the side work reads/writes -- held until join_side() so the caching
    allocator cannot hand their memory to later main-stream allocations."""

    def __init__(self, device, keep=(), lane=1, after=None):
        self.dev = device.index or 0
        self.keep = list(keep)
        self.lane = int(lane)
        # after: an event already recorded on the main stream (stream_event()): the lane starts behind THAT point and
        # nothing new is put on the main stream (every marker there costs ~6 us of an in-order queue's time: measured
        # 25 us between a layer's dx GEMM and the next BPTT kernel with four lane entries in between)
        self.after = after

    def __enter__(self):
        global _lane
        st = _side.get((self.dev, self.lane))
        if st is None:
            st = _side[(self.dev, self.lane)] = dict(stream=torch.cuda.Stream(device=self.dev), keep=[])
        st['keep'].extend(self.keep)
        if self.after is not None:
            st['stream'].wait_event(self.after)
        else:
            st['stream'].wait_stream(_cur_stream(self.dev))
        self._ctx = torch.cuda.stream(st['stream'])
        self._ctx.__enter__()
        self._prev = _lane
        _lane = self.lane
        return self

    def __exit__(self, *exc):
        global _lane
        _lane = self._prev
        self._ctx.__exit__(*exc)
        return False


def set_side_xcd_skip(device, n):
    """The side lanes' reduction-major GEMMs leave the first n XCDs alone (asr_set_xcd_skip).  An experiment knob, off
    by default: keeping the weight-gradient GEMMs off the recurrence clusters' XCDs (their L2s carry the per-step
    hand-off) was measured on the headline step -- BPTT launch 1321 us (n = 0), 1314 us (n = 2), 1362 us (n = 4) -- so
    whatever stretches the recurrence beside the side GEMMs (1.19 ms alone) is not L2 sharing."""
    dev = device.index or 0
    for lane in (1, 2):
        h = _lib.handle(dev, lane)
        if getattr(h, '_xcd_skip', None) != n:
            h.check(h.lib.asr_set_xcd_skip(h.h, int(n)), 'asr_set_xcd_skip')
            h._xcd_skip = n


def set_side_gemm_workgroups(device, n):
    """Workgroups the side lanes' reduction-major (weight-gradient) GEMMs aim at (asr_set_gemm_tn_workgroups): 32 for
    GEMMs that run beside a recurrence kernel, 0 (default, ~2 per CU) for exposed ones."""
    dev = device.index or 0
    for lane in (1, 2):
        h = _lib.handle(dev, lane)
        if getattr(h, '_tn_wgs', None) != n:
            h.check(h.lib.asr_set_gemm_tn_workgroups(h.h, int(n)), 'asr_set_gemm_tn_workgroups')
            h._tn_wgs = n


def keep_on_lane(device, lane, tensors):
    """Hold `tensors` until join_side(): work already issued on side lane `lane` reads or writes them."""
    st = _side.get((device.index or 0, int(lane)))
    if st is not None:
        st['keep'].extend(tensors)


def join_side(device):
    """Main stream waits for all side-lane work; releases the tensors held for it."""
    dev = device.index or 0
    for (d, lane), st in _side.items():
        if d != dev:
            continue
        _cur_stream(dev).wait_stream(st['stream'])
        st['keep'] = []


def stream_event():
    """Event recorded on the CURRENT stream (inside a side_lane block: on that lane).  wait_event(ev) makes the
    then-current stream wait for exactly this point instead of for everything on the other stream."""
    ev = torch.cuda.Event()
    ev.record(_cur_stream())
    return ev


def wait_event(ev):
    if ev is not None:
        _cur_stream().wait_event(ev)


_stream_objs = {}


def _raw_stream(dev=None):
    """Raw hipStream_t of torch's current stream (torch.cuda.current_stream() resolves the device through four Python
    layers every time: 1.1 ms of a 3.8 ms step issue at ~160 calls per step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if dev is None else dev)


def _cur_stream(dev=None):
    """torch.cuda.current_stream(dev) as a cached Stream object, looked up by its raw handle."""
    # the default stream's raw handle is 0 on EVERY device: the key must carry the resolved device index, or a process
    # that touches a second GPU would be handed the first one's stream object
    d = torch._C._cuda_getDevice() if dev is None else (dev.index if isinstance(dev, torch.device) else dev)
    if d is None:
        d = torch._C._cuda_getDevice()
    raw = torch._C._cuda_getCurrentRawStream(d)
    key = (d, raw)
    so = _stream_objs.get(key)
    if so is None:
        so = _stream_objs[key] = torch.cuda.current_stream(dev)
    return so


def _s():
    return C.c_void_p(_raw_stream())


def to_device(x, dtype, dev):
    """Host array / list / CPU tensor -> device tensor WITHOUT draining the stream: staged in pinned memory and copied
    with non_blocking=True (a pageable `torch.as_tensor(x, device=dev)` waits for everything enqueued before it -- one
    such call per training step makes the whole step host-synchronous).  Device tensors pass through (dtype converted on
    the device).  torch's caching host allocator keeps the pinned block alive until the copy has run."""
    if torch.is_tensor(x) and x.is_cuda:
        return x if x.dtype == dtype else x.to(dtype)
    t = torch.as_tensor(x) if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.contiguous()
    if torch.device(dev).type != 'cuda':
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


_upload_streams = {}


def upload_ints(dev, vectors):
    """Host int vectors -> int32 device vectors through ONE pinned staging buffer and ONE asynchronous copy, issued on
    an upload stream of its own and joined to the current stream by an event.  A CTC step hands over three such vectors
    (frame counts, label offsets, labels); as three copies on the main stream they and the queue gaps between them held
    the first kernel of the step back by 50 us behind the previous step's optimizer (profiles/r06_step_timeline.md); on
    the upload stream the copy runs as soon as the host has issued it, beside whatever the main stream is still doing.
    (Step time of the headline bench is unchanged by it, 8.90 ms: without the profiler the host runs steps ahead and the
    copies' gaps were the profiler's; what remains is two packets fewer on the main stream.)
    Device vectors pass through (converted to int32 on the device).  Each result starts on a 16-byte boundary."""
    d = torch.device(dev)
    out, host = [None] * len(vectors), []
    for i, v in enumerate(vectors):
        if torch.is_tensor(v) and v.is_cuda:
            out[i] = v if v.dtype == torch.int32 else v.to(torch.int32)
        else:
            a = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
            host.append((i, np.ascontiguousarray(a, dtype=np.int32).ravel()))
    if not host:
        return out
    if d.type != 'cuda':
        for i, a in host:
            out[i] = torch.from_numpy(a.copy()).to(d)
        return out
    starts, n = [], 0
    for _, a in host:
        starts.append(n)
        n += max(4, a.size + (-a.size) % 4)
    stage = torch.zeros(n, dtype=torch.int32).pin_memory()
    sv = stage.numpy()
    for (_, a), s0 in zip(host, starts):
        sv[s0:s0 + a.size] = a
    idx = d.index if d.index is not None else torch._C._cuda_getDevice()
    up = _upload_streams.get(idx)
    if up is None:
        up = _upload_streams[idx] = torch.cuda.Stream(device=idx)
    with torch.cuda.stream(up):
        buf = stage.to(d, non_blocking=True)      # allocated in the upload stream's pool: no wait for the main stream
    ev = torch.cuda.Event()
    ev.record(up)
    cur = _cur_stream(idx)
    cur.wait_event(ev)
    buf.record_stream(cur)                        # side lanes fork from / join into this stream within the step
    for (i, a), s0 in zip(host, starts):
        out[i] = buf[s0:s0 + a.size]
    return out


_host_copies = {}


def host_ints(x):
    """Host numpy view of an integer vector the caller may hand over on the host (the reference feeds seq_len as numpy,
    examples/timit/training/train_ctc.py:134-144) or on the device.  A device tensor costs ONE synchronising copy the
    first time it is seen; the copy is remembered for that tensor OBJECT (weak reference + version counter: a new tensor
    that merely reuses the address, or an in-place update, is read again), so a batch object reused across steps (bench
    loops, an epoch cached on the device) never drains the stream a second time."""
    if not torch.is_tensor(x):
        return np.asarray(x)
    if not x.is_cuda:
        return x.detach().numpy()
    import weakref
    hit = _host_copies.get(id(x))
    if hit is not None and hit[0]() is x and hit[1] == x._version:
        return hit[2]
    if len(_host_copies) > 64:
        for k in [k for k, v in _host_copies.items() if v[0]() is None]:
            del _host_copies[k]
        if len(_host_copies) > 64:
            _host_copies.clear()
    host = x.detach().cpu().numpy()
    _host_copies[id(x)] = (weakref.ref(x), x._version, host)
    return host


def invalidate_host_ints(x=None):
    """Forget the remembered host copy of device vector `x` (all of them when None).  The version counter host_ints keys
    on does not move for writes torch does not see -- `x.data.copy_()`, a kernel launched on `x.data_ptr()` through ctypes
    (this library's own calls), DLPack consumers: a caller that rewrites a length vector that way, in place, says so here
    (or, simpler, hands the lengths over on the host, as the reference's feed_dict does)."""
    if x is None:
        _host_copies.clear()
    else:
        _host_copies.pop(id(x), None)


def _p(t):
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t.dtype != dtype:
        raise ValueError('%s must be %s, got %s' % (name, dtype, t.dtype))


# ---------------------------------------------------------------- layout / casts
def bt_to_tb(x_btd, dtype=ASR_F32, ld=None):
    """[B,T,D] fp32 -> [T,B,ld] in `dtype`; columns D..ld-1 (ld defaults to D) are zero -- the reduction width the
    lean GEMM wants (a multiple of 64) when D is not one."""
    h = _h(x_btd)
    _chk(x_btd, torch.float32, 'inputs')
    B, T, D = x_btd.shape
    ld = D if ld is None else int(ld)
    out = torch.empty((T, B, ld), dtype=TORCH_DTYPE[dtype], device=x_btd.device)
    h.check(h.lib.asr_bt_to_tb_ld(h.h, dtype, _p(x_btd), _p(out), B, T, D, ld, _s()), 'asr_bt_to_tb')
    return out


def stack_frames(x_btf, seq_len, num_stack, num_skip):
    """Zero-padded batch [B,T,F] fp32 + seq_len [B] int32 -> ([B, ceil(T/num_skip), F*num_stack], new seq_len)."""
    h = _h(x_btf)
    _chk(x_btf, torch.float32, 'inputs')
    _chk(seq_len, torch.int32, 'seq_len')
    if num_stack < num_skip:
        raise ValueError('num_skip must be less than num_stack.')
    B, T, F = x_btf.shape
    Tn = (T + num_skip - 1) // num_skip
    out = torch.empty((B, Tn, F * num_stack), dtype=torch.float32, device=x_btf.device)
    out_len = torch.empty((B,), dtype=torch.int32, device=x_btf.device)
    h.check(h.lib.asr_stack_frames(h.h, _p(x_btf), _p(seq_len), B, T, F, int(num_stack), int(num_skip), _p(out),
                                   _p(out_len), _s()), 'asr_stack_frames')
    return out, out_len


def splice(x_btd, seq_len, splice, num_stack=1):
    """Zero-padded batch [B,T,D] fp32 -> [B,T,D*splice], spliced per utterance over its own seq_len frames."""
    h = _h(x_btd)
    _chk(x_btd, torch.float32, 'inputs')
    _chk(seq_len, torch.int32, 'seq_len')
    B, T, D = x_btd.shape
    if D % (3 * num_stack) != 0:
        raise ValueError('frame width must be channels*3*num_stack')
    out = torch.empty((B, T, D * splice), dtype=torch.float32, device=x_btd.device)
    h.check(h.lib.asr_splice(h.h, _p(x_btd), _p(seq_len), B, T, D, int(splice), int(num_stack), _p(out), _s()),
            'asr_splice')
    return out


def transpose2d(x, out=None):
    """out[c, r] = x[r, c] for a 2-D tensor with unit inner stride."""
    h = _h(x)
    dt = dtype_id(x.dtype)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError('transpose2d: 2-D tensor with unit inner stride expected')
    R, Cc = x.shape
    if out is None:
        out = torch.empty((Cc, R), dtype=x.dtype, device=x.device)
    h.check(h.lib.asr_transpose2d(h.h, dt, C.c_void_p(x.data_ptr()), R, Cc, x.stride(0),
                                  C.c_void_p(out.data_ptr()), out.stride(0), _s()), 'asr_transpose2d')
    return out


def cast_from_f32(x, dtype, out=None):
    h = _h(x)
    _chk(x, torch.float32, 'x')
    if out is None:
        out = torch.empty(x.shape, dtype=TORCH_DTYPE[dtype], device=x.device)
    h.check(h.lib.asr_cast_from_f32(h.h, dtype, _p(x), _p(out), x.numel(), _s()), 'asr_cast_from_f32')
    return out


def cast_to_f32(x, out=None):
    h = _h(x)
    dt = dtype_id(x.dtype)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    h.check(h.lib.asr_cast_to_f32(h.h, dt, _p(x), _p(out), x.numel(), _s()), 'asr_cast_to_f32')
    return out


def apply_mask(x, mask, out=None):
    h = _h(x)
    dt = dtype_id(x.dtype)
    _chk(mask, torch.float32, 'mask')
    if out is None:
        out = torch.empty_like(x)
    h.check(h.lib.asr_apply_mask(h.h, dt, _p(x), _p(mask), _p(out), x.numel(), _s()), 'asr_apply_mask')
    return out


def dropout_mask(shape, keep_prob, seed, offset, device):
    mask = torch.empty(shape, dtype=torch.float32, device=device)
    h = _h(mask)
    h.check(h.lib.asr_dropout_mask(h.h, _p(mask), mask.numel(), float(keep_prob), int(seed),
                                   int(offset), _s()), 'asr_dropout_mask')
    return mask


def touch(t):
    """Read pass over a tensor (asr_touch): into the memory-side cache ahead of a latency-critical consumer."""
    h = _h(t)
    h.check(h.lib.asr_touch(h.h, _p(t), t.numel() * t.element_size(), _s()), 'asr_touch')


def dropout_apply(x, keep_prob, seed, offset):
    """x * dropout mask(seed, offset) without a mask tensor (== apply_mask(x, dropout_mask(x.shape, ...)) bit for bit)."""
    h = _h(x)
    out = torch.empty_like(x)
    h.check(h.lib.asr_dropout_apply(h.h, dtype_id(x.dtype), _p(x), _p(out), x.numel(), float(keep_prob), int(seed),
                                    int(offset), _s()), 'asr_dropout_apply')
    return out


def colsum(a, out=None):
    """sum over rows of a [M,N] (last-dim contiguous 2-D view) -> fp32 [N]."""
    h = _h(a)
    dt = dtype_id(a.dtype)
    M, N = a.shape
    if a.stride(1) != 1:
        raise ValueError('colsum: inner stride must be 1')
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=a.device)
    h.check(h.lib.asr_colsum(h.h, dt, C.c_void_p(a.data_ptr()), M, N, a.stride(0), _p(out), _s()),
            'asr_colsum')
    return out


# ---------------------------------------------------------------- GEMM
def gemm(A, B, transA=False, transB=False, bias=None, out=None, out_dtype=None, accumulate=False, relu=False,
         mul=None, drop=None):
    """C = op(A) @ op(B) (+bias) (+C).  A, B 2-D, same dtype (f32 or bf16), row stride = ld.
    mul: fp32 [M,N] multiplier of the result (a dropout mask); drop = (keep_prob, seed, offset): the same multiplier
    formed from the dropout generator's counter in the epilogue -- dropout_mask((M, N), ...) without the tensor."""
    h = _h(A)
    dt = dtype_id(A.dtype)
    if B.dtype != A.dtype:
        raise ValueError('gemm: A and B dtypes differ')
    for t, n in ((A, 'A'), (B, 'B')):
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError('gemm: %s must be 2-D with unit inner stride' % n)
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    Kb, N = (B.shape[1], B.shape[0]) if transB else (B.shape[0], B.shape[1])
    if K != Kb:
        raise ValueError('gemm: inner dimensions differ (%d vs %d)' % (K, Kb))
    odt = dt if out_dtype is None else dtype_id(out_dtype)
    if out is None:
        out = torch.empty((M, N), dtype=TORCH_DTYPE[odt], device=A.device)
    else:
        odt = dtype_id(out.dtype)
        if out.shape != (M, N) or out.stride(1) != 1:
            raise ValueError('gemm: bad out shape %s' % (tuple(out.shape),))
    if bias is not None:
        _chk(bias, torch.float32, 'bias')
    if drop is not None:
        if odt != ASR_F32 or not out.is_contiguous() or N % 4:
            raise ValueError('gemm: drop needs a contiguous fp32 output with N % 4 == 0')
        h.check(h.lib.asr_gemm_drop(h.h, dt, int(transA), int(transB), M, N, K,
                                    C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()), B.stride(0),
                                    C.c_void_p(out.data_ptr()), out.stride(0), _p(bias), int(accumulate),
                                    1 if relu else 0, float(drop[0]), int(drop[1]), int(drop[2]), _s()), 'asr_gemm_drop')
        return out
    if mul is not None:   # fp32 output times an elementwise fp32 multiplier [M,N] (a dropout mask), in the epilogue
        _chk(mul, torch.float32, 'mul')
        if odt != ASR_F32 or mul.dim() != 2 or mul.shape != (M, N) or mul.stride(1) != 1:
            raise ValueError('gemm: mul needs an fp32 output and an fp32 [M,N] multiplier')
        h.check(h.lib.asr_gemm_mul(h.h, dt, int(transA), int(transB), M, N, K,
                                   C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()), B.stride(0),
                                   C.c_void_p(out.data_ptr()), out.stride(0), _p(bias), int(accumulate),
                                   1 if relu else 0, C.c_void_p(mul.data_ptr()), mul.stride(0), _s()), 'asr_gemm_mul')
        return out
    h.check(h.lib.asr_gemm_act(h.h, dt, odt, int(transA), int(transB), M, N, K,
                               C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()), B.stride(0),
                               C.c_void_p(out.data_ptr()), out.stride(0), _p(bias), int(accumulate),
                               1 if relu else 0, _s()), 'asr_gemm')
    return out


# ---------------------------------------------------------------- VGG front-end
def conv3x3_prep_weights(w_hwio):
    """fp32 [3,3,Cin,Cout] -> (wt_fwd bf16 [Cout, 9*Cin], wt_bwd bf16 [Cin, 9*Cout])."""
    h = _h(w_hwio)
    _chk(w_hwio, torch.float32, 'w')
    _, _, Cin, Cout = w_hwio.shape
    wf = torch.empty((Cout, 9 * Cin), dtype=torch.bfloat16, device=w_hwio.device)
    wb = torch.empty((Cin, 9 * Cout), dtype=torch.bfloat16, device=w_hwio.device)
    h.check(h.lib.asr_conv3x3_prep_weights(h.h, _p(w_hwio.contiguous()), Cin, Cout, _p(wf), _p(wb), _s()),
            'asr_conv3x3_prep_weights')
    return wf, wb


def _out_like(out, shape, dtype, device, what):
    """`out` (a contiguous tensor of exactly this shape / dtype, e.g. a slice of a larger one along the first axis) or a fresh one."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or not out.is_contiguous():
        raise ValueError('%s: out must be a contiguous %s tensor of shape %s' % (what, dtype, tuple(shape)))
    return out


def conv3x3_fwd(x_nhwc, wt_fwd, bias, relu=True, out=None):
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    N, H, W, Cin = x_nhwc.shape
    Cout = wt_fwd.shape[0]
    out = _out_like(out, (N, H, W, Cout), torch.bfloat16, x_nhwc.device, 'conv3x3_fwd')
    h.check(h.lib.asr_conv3x3_fwd(h.h, _p(x_nhwc), N, H, W, Cin, _p(wt_fwd), _p(bias), Cout, 1 if relu else 0,
                                  _p(out), _s()), 'asr_conv3x3_fwd')
    return out


def conv3x3_fwd_drop(x_nhwc, wt_fwd, bias, drop, out=None):
    """dropout_apply(conv3x3_fwd(x, ..., relu=True), *drop) in one launch (the undropped activation is never written)."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    N, H, W, Cin = x_nhwc.shape
    Cout = wt_fwd.shape[0]
    out = _out_like(out, (N, H, W, Cout), torch.bfloat16, x_nhwc.device, 'conv3x3_fwd_drop')
    h.check(h.lib.asr_conv3x3_fwd_drop(h.h, _p(x_nhwc), N, H, W, Cin, _p(wt_fwd), _p(bias), Cout, float(drop[0]),
                                       int(drop[1]), int(drop[2]), _p(out), _s()), 'asr_conv3x3_fwd_drop')
    return out


def conv3x3_bwd_data(dy_nhwc, wt_bwd):
    h = _h(dy_nhwc)
    _chk(dy_nhwc, torch.bfloat16, 'dy')
    N, H, W, Cout = dy_nhwc.shape
    Cin = wt_bwd.shape[0]
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy_nhwc.device)
    h.check(h.lib.asr_conv3x3_bwd_data(h.h, _p(dy_nhwc), N, H, W, Cout, _p(wt_bwd), Cin, _p(dx), _s()),
            'asr_conv3x3_bwd_data')
    return dx


def conv3x3_smallc_fwd(x_nhwc, w2d, bias, relu=True):
    """Direct 3x3 SAME convolution for 9*Cin <= 32, Cout == 64 (bf16): x [N,H,W,Cin], w2d [9*Cin, 64] bf16 -> [N,H,W,64]."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(w2d, torch.bfloat16, 'w2d')
    N, H, W, Cin = x_nhwc.shape
    out = torch.empty((N, H, W, w2d.shape[1]), dtype=torch.bfloat16, device=x_nhwc.device)
    h.check(h.lib.asr_conv3x3_smallc_fwd(h.h, _p(x_nhwc), N, H, W, Cin, _p(w2d), _p(bias), w2d.shape[1], int(relu),
                                         _p(out), _s()), 'asr_conv3x3_smallc_fwd')
    return out


def conv3x3_smallc_fwd_drop(x_nhwc, w2d, bias, drop, out=None):
    """dropout_apply(conv3x3_smallc_fwd(x, ..., relu=True), *drop) in one launch."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(w2d, torch.bfloat16, 'w2d')
    N, H, W, Cin = x_nhwc.shape
    out = _out_like(out, (N, H, W, w2d.shape[1]), torch.bfloat16, x_nhwc.device, 'conv3x3_smallc_fwd_drop')
    h.check(h.lib.asr_conv3x3_smallc_fwd_drop(h.h, _p(x_nhwc), N, H, W, Cin, _p(w2d), _p(bias), w2d.shape[1],
                                              float(drop[0]), int(drop[1]), int(drop[2]), _p(out), _s()),
            'asr_conv3x3_smallc_fwd_drop')
    return out


def conv3x3_smallc_bwd_weight(x_nhwc, dpre_nhwc, dw):
    """dw fp32 [9*Cin, 64] = patches(x)^T dpre (both bf16), no patch matrix."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(dpre_nhwc, torch.bfloat16, 'dpre')
    N, H, W, Cin = x_nhwc.shape
    h.check(h.lib.asr_conv3x3_smallc_bwd_weight(h.h, _p(x_nhwc), _p(dpre_nhwc), N, H, W, Cin, dpre_nhwc.shape[-1],
                                                _p(dw), _s()), 'asr_conv3x3_smallc_bwd_weight')
    return dw


def conv3x3_smallc_bwd_weight_bias(x_nhwc, dpre_nhwc, dw, dbias):
    """conv3x3_smallc_bwd_weight + dbias fp32 [64] = column sums of dpre, in the same two launches."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(dpre_nhwc, torch.bfloat16, 'dpre')
    _chk(dbias, torch.float32, 'dbias')
    N, H, W, Cin = x_nhwc.shape
    if dbias.numel() != dpre_nhwc.shape[-1]:
        raise ValueError('conv3x3_smallc_bwd_weight_bias: dbias has %d elements' % dbias.numel())
    h.check(h.lib.asr_conv3x3_smallc_bwd_weight_bias(h.h, _p(x_nhwc), _p(dpre_nhwc), N, H, W, Cin, dpre_nhwc.shape[-1],
                                                     _p(dw), _p(dbias), _s()), 'asr_conv3x3_smallc_bwd_weight_bias')
    return dw, dbias


def conv3x3_bwd_data_relu(dy_nhwc, wt_bwd, act_below, drop=None, dropped=False):
    """relu_bwd(conv3x3_bwd_data(dy, wt_bwd), act_below, drop=drop) without the fp32 gradient in between -> bf16."""
    h = _h(dy_nhwc)
    _chk(dy_nhwc, torch.bfloat16, 'dy')
    _chk(act_below, torch.bfloat16, 'act_below')
    N, H, W, Cout = dy_nhwc.shape
    Cin = wt_bwd.shape[0]
    dpre = torch.empty((N, H, W, Cin), dtype=torch.bfloat16, device=dy_nhwc.device)
    k, sd, off = drop if drop is not None else (1.0, 0, 0)
    # dropped=True: act_below is the DROPPED activation (conv3x3_fwd_drop): no mask is formed, dx * (1 / keep) where it is > 0
    mode = 0 if drop is None else (2 if dropped else 1)
    h.check(h.lib.asr_conv3x3_bwd_data_relu(h.h, _p(dy_nhwc), N, H, W, Cout, _p(wt_bwd), Cin, _p(act_below), float(k),
                                            int(sd), int(off), mode, _p(dpre), _s()),
            'asr_conv3x3_bwd_data_relu')
    return dpre


def conv3x3_bwd_weight(x_nhwc, dy_nhwc, dw, accumulate=False):
    """dw: fp32 [9*Cin, Cout] view of the HWIO gradient."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(dy_nhwc, torch.bfloat16, 'dy')
    N, H, W, Cin = x_nhwc.shape
    Cout = dy_nhwc.shape[3]
    h.check(h.lib.asr_conv3x3_bwd_weight(h.h, _p(x_nhwc), _p(dy_nhwc), N, H, W, Cin, Cout, _p(dw),
                                         int(accumulate), _s()), 'asr_conv3x3_bwd_weight')
    return dw


def conv3x3_bwd_weight_bias(x_nhwc, dy_nhwc, dw, dbias):
    """dw: fp32 [9*Cin, Cout] view of the HWIO gradient, dbias: fp32 [Cout] = column sums of dy -- one call, the bias sums
    from the dy images the weight-gradient kernel stages anyway (asr_conv3x3_bwd_weight_bias)."""
    h = _h(x_nhwc)
    _chk(x_nhwc, torch.bfloat16, 'x')
    _chk(dy_nhwc, torch.bfloat16, 'dy')
    _chk(dbias, torch.float32, 'dbias')
    N, H, W, Cin = x_nhwc.shape
    Cout = dy_nhwc.shape[3]
    if dbias.numel() != Cout:
        raise ValueError('conv3x3_bwd_weight_bias: dbias has %d elements, Cout = %d' % (dbias.numel(), Cout))
    h.check(h.lib.asr_conv3x3_bwd_weight_bias(h.h, _p(x_nhwc), _p(dy_nhwc), N, H, W, Cin, Cout, _p(dw), _p(dbias), _s()),
            'asr_conv3x3_bwd_weight_bias')
    return dw, dbias


def im2col3x3(x_nhwc, ldp=None, out=None):
    h = _h(x_nhwc)
    dt = dtype_id(x_nhwc.dtype)
    N, H, W, Cin = x_nhwc.shape
    ldp = ldp or 9 * Cin
    if out is None:
        out = torch.zeros((N * H * W, ldp), dtype=x_nhwc.dtype, device=x_nhwc.device)
    h.check(h.lib.asr_im2col3x3(h.h, dt, _p(x_nhwc), N, H, W, Cin, ldp, _p(out), _s()), 'asr_im2col3x3')
    return out


def col2im3x3(dpatches, N, H, W, Cin):
    h = _h(dpatches)
    _chk(dpatches, torch.float32, 'dpatches')
    din = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dpatches.device)
    h.check(h.lib.asr_col2im3x3(h.h, _p(dpatches), N, H, W, Cin, dpatches.stride(0), _p(din), _s()), 'asr_col2im3x3')
    return din


def conv_out_hw(H, W, sh, sw):
    return (H + sh - 1) // sh, (W + sw - 1) // sw


def im2col(x_nhwc, kh, kw, sh, sw, ldp=None):
    """[N,H,W,Cin] -> patches [N*Ho*Wo, ldp] of a SAME convolution with any kernel / stride (columns past
    kh*kw*Cin zero)."""
    h = _h(x_nhwc)
    dt = dtype_id(x_nhwc.dtype)
    N, H, W, Cin = x_nhwc.shape
    Ho, Wo = conv_out_hw(H, W, sh, sw)
    K = kh * kw * Cin
    ldp = ldp or K
    out = torch.zeros((N * Ho * Wo, ldp), dtype=x_nhwc.dtype, device=x_nhwc.device) if ldp > K else \
        torch.empty((N * Ho * Wo, ldp), dtype=x_nhwc.dtype, device=x_nhwc.device)
    h.check(h.lib.asr_im2col(h.h, dt, _p(x_nhwc), N, H, W, Cin, kh, kw, sh, sw, ldp, _p(out), _s()), 'asr_im2col')
    return out


def col2im(dpatches, N, H, W, Cin, kh, kw, sh, sw):
    h = _h(dpatches)
    _chk(dpatches, torch.float32, 'dpatches')
    din = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dpatches.device)
    h.check(h.lib.asr_col2im(h.h, _p(dpatches), N, H, W, Cin, kh, kw, sh, sw, dpatches.stride(0), _p(din), _s()),
            'asr_col2im')
    return din


def maxpool2x2_fwd(x_nhwc):
    h = _h(x_nhwc)
    dt = dtype_id(x_nhwc.dtype)
    N, H, W, Cc = x_nhwc.shape
    out = torch.empty((N, (H + 1) // 2, (W + 1) // 2, Cc), dtype=x_nhwc.dtype, device=x_nhwc.device)
    arg = torch.empty(out.shape, dtype=torch.uint8, device=x_nhwc.device)
    h.check(h.lib.asr_maxpool2x2_fwd(h.h, dt, _p(x_nhwc), N, H, W, Cc, _p(out), _p(arg), _s()), 'asr_maxpool2x2_fwd')
    return out, arg


def maxpool2x2_fwd_drop(x_nhwc, drop, out=None, arg=None):
    """dropout_apply(maxpool2x2_fwd(x)[0], *drop) and the argmax in one launch."""
    h = _h(x_nhwc)
    dt = dtype_id(x_nhwc.dtype)
    N, H, W, Cc = x_nhwc.shape
    out = _out_like(out, (N, (H + 1) // 2, (W + 1) // 2, Cc), x_nhwc.dtype, x_nhwc.device, 'maxpool2x2_fwd_drop')
    arg = _out_like(arg, tuple(out.shape), torch.uint8, x_nhwc.device, 'maxpool2x2_fwd_drop(arg)')
    h.check(h.lib.asr_maxpool2x2_fwd_drop(h.h, dt, _p(x_nhwc), N, H, W, Cc, _p(out), _p(arg), float(drop[0]), int(drop[1]),
                                          int(drop[2]), _s()), 'asr_maxpool2x2_fwd_drop')
    return out, arg


def maxpool2x2_bwd(dout, arg, H, W):
    h = _h(dout)
    N, _, _, Cc = dout.shape
    din = torch.empty((N, H, W, Cc), dtype=torch.float32, device=dout.device)
    h.check(h.lib.asr_maxpool2x2_bwd(h.h, _p(dout), _p(arg), N, H, W, Cc, _p(din), _s()), 'asr_maxpool2x2_bwd')
    return din


def maxpool2x2_relu_bwd(dout, arg, act, drop=None, pooled=None, hw=None):
    """relu_bwd(maxpool2x2_bwd(dropout_apply(dout, *drop)), act) in one pass: dout [N,Ho,Wo,C] fp32 pooled gradient,
    arg the pool's argmax, act [N,H,W,C] the ReLU output under the pool (operand dtype) -> dpre like act.
    pooled (with hw = (H, W)): the POOLED activation after its dropout (maxpool2x2_fwd_drop; or the plain pooled output when
    drop is None) instead of act -- it is > 0 exactly where the window's maximum was active and kept, so neither the
    full-resolution activation is read nor a mask formed."""
    h = _h(dout)
    k, sd, off = drop if drop is not None else (1.0, 0, 0)
    if pooled is not None:
        N, _, _, Cc = pooled.shape
        H, W = hw
        dpre = torch.empty((N, H, W, Cc), dtype=pooled.dtype, device=pooled.device)
        h.check(h.lib.asr_maxpool2x2_relu_bwd(h.h, dtype_id(pooled.dtype), _p(dout), _p(arg), _p(pooled), N, H, W, Cc, _p(dpre),
                                              float(k), 0, 0, 2, _s()), 'asr_maxpool2x2_relu_bwd')
        return dpre
    N, H, W, Cc = act.shape
    dpre = torch.empty_like(act)
    h.check(h.lib.asr_maxpool2x2_relu_bwd(h.h, dtype_id(act.dtype), _p(dout), _p(arg), _p(act), N, H, W, Cc, _p(dpre),
                                          float(k), int(sd), int(off), int(drop is not None), _s()),
            'asr_maxpool2x2_relu_bwd')
    return dpre


def relu_bwd_scaled(dout, out_dropped, keep):
    """dpre = (out_dropped > 0) ? dout * (1 / keep) : 0 (out_dropped: a ReLU output after its dropout with keep_prob keep)."""
    h = _h(dout)
    dpre = torch.empty_like(out_dropped)
    h.check(h.lib.asr_relu_bwd_scaled(h.h, dtype_id(out_dropped.dtype), _p(dout), _p(out_dropped), out_dropped.numel(),
                                      float(keep), _p(dpre), _s()), 'asr_relu_bwd_scaled')
    return dpre


def relu_bwd(dout, out, mask=None, drop=None):
    """dout fp32, out in the operand dtype -> dpre (operand dtype) = dout * (out>0) (* mask).
    drop = (keep_prob, seed, offset): the mask of dropout_apply with those arguments, formed in the kernel."""
    h = _h(dout)
    dt = dtype_id(out.dtype)
    dpre = torch.empty_like(out)
    if drop is not None:
        h.check(h.lib.asr_relu_bwd_drop(h.h, dt, _p(dout), _p(out), out.numel(), float(drop[0]), int(drop[1]),
                                        int(drop[2]), _p(dpre), _s()), 'asr_relu_bwd_drop')
        return dpre
    h.check(h.lib.asr_relu_bwd(h.h, dt, _p(dout), _p(out), _p(mask), out.numel(), _p(dpre), _s()), 'asr_relu_bwd')
    return dpre


# ---------------------------------------------------------------- LSTM
def lstm_prep_weights(kernel, bias, din, H, dtype, out=None):
    """kernel [Din+H,4H] fp32 (TF layout), bias [4H] -> dict(wx_il [Din,4H] dtype, bias_il [4H] fp32,
    pf, pb packed W_h fragments)."""
    h = _h(kernel)
    _chk(kernel, torch.float32, 'kernel')
    if kernel.shape != (din + H, 4 * H):
        raise ValueError('kernel must be [Din+H,4H]')
    dev, tdt = kernel.device, TORCH_DTYPE[dtype]
    if out is None:
        out = dict(wx_il=torch.empty((din, 4 * H), dtype=tdt, device=dev),
                   bias_il=torch.empty((4 * H,), dtype=torch.float32, device=dev),
                   pf=torch.empty((H * 4 * H,), dtype=tdt, device=dev),
                   pb=torch.empty((H * 4 * H,), dtype=tdt, device=dev))
    h.check(h.lib.asr_lstm_prep_weights(h.h, dtype, _p(kernel), _p(bias), din, H, _p(out['wx_il']),
                                        _p(out['bias_il']), _p(out['pf']), _p(out['pb']), _s()),
            'asr_lstm_prep_weights')
    return out


def gate_deinterleave(src, dst, H):
    """src [R,4H] fp32 interleaved columns -> dst [R,4H] gate-major (row strides honoured)."""
    h = _h(src)
    h.check(h.lib.asr_gate_deinterleave(h.h, C.c_void_p(src.data_ptr()), src.stride(0),
                                        C.c_void_p(dst.data_ptr()), dst.stride(0), src.shape[0], H, _s()),
            'asr_gate_deinterleave')
    return dst


def _ptr_table(rows):
    """ndir x 5 device pointers (kernel, bias, w_i_diag, w_f_diag, w_o_diag per direction) as a C array."""
    flat = []
    for r in rows:
        r = list(r) + [None] * (5 - len(r))
        flat.extend(t.data_ptr() if t is not None else None for t in r)
    return (C.c_void_p * len(flat))(*flat)


def lstm_prep_layer(variables, din, H, dtype, ldk=None):
    """variables: per direction (kernel [Din+H,4H], bias [4H], w_i_diag, w_f_diag, w_o_diag [H] or None) fp32 views.
    One launch -> dict(wxT [ndir*4H, ldk], wx_cat [Din, ndir*4H], bias [ndir*4H] fp32, whf / whb [ndir, H*4H] packed
    W_h, peep [ndir,3,H] fp32 or None) in the operand dtype (include/asr_hip.h asr_lstm_prep_layer)."""
    k0 = variables[0][0]
    h = _h(k0)
    ndir = len(variables)
    ldk = din if ldk is None else int(ldk)
    for v in variables:
        _chk(v[0], torch.float32, 'kernel')
        if v[0].shape != (din + H, 4 * H):
            raise ValueError('kernel must be [Din+H,4H]')
    has_peep = len(variables[0]) >= 5 and variables[0][2] is not None
    dev, tdt = k0.device, TORCH_DTYPE[dtype]
    out = dict(wxT=torch.empty((ndir * 4 * H, ldk), dtype=tdt, device=dev),
               wx_cat=torch.empty((din, ndir * 4 * H), dtype=tdt, device=dev),
               bias=torch.empty((ndir * 4 * H,), dtype=torch.float32, device=dev),
               whf=torch.empty((ndir, H * 4 * H), dtype=tdt, device=dev),
               whb=torch.empty((ndir, H * 4 * H), dtype=tdt, device=dev),
               peep=torch.empty((ndir, 3, H), dtype=torch.float32, device=dev) if has_peep else None)
    h.check(h.lib.asr_lstm_prep_layer(h.h, dtype, ndir, _ptr_table(variables), din, ldk, H, _p(out['wxT']),
                                      _p(out['wx_cat']), _p(out['bias']), _p(out['whf']), _p(out['whb']),
                                      _p(out['peep']), _s()), 'asr_lstm_prep_layer')
    return out


def lstm_grad_finish(grads, dw_il, dpeep, H):
    """dw_il [ndir, Din+H, 4H] fp32 (interleaved columns) + dpeep [ndir,7,H] -> the gradient views `grads`
    (per direction: kernel, bias, w_i_diag, w_f_diag, w_o_diag or None), TF layouts, one launch."""
    h = _h(dw_il)
    ndir, rows, _ = dw_il.shape
    has_peep = len(grads[0]) >= 5 and grads[0][2] is not None
    h.check(h.lib.asr_lstm_grad_finish(h.h, ndir, _ptr_table(grads), rows, H, _p(dw_il), _p(dpeep),
                                       1 if has_peep else 0, _s()), 'asr_lstm_grad_finish')


LSTM_UNITS = (64, 128, 192, 256, 320, 512)   # include/asr_hip.h asr_lstm_fwd


def lstm_units_supported(H):
    """Whether asr_lstm_fwd / asr_lstm_bwd take num_units = H (callers with a step-by-step alternative ask first)."""
    return int(H) in LSTM_UNITS


def lstm_fwd(xproj, wh_packed, peep, seq_len, H, ndir, dtype, forget_bias=1.0, cell_clip=0.0,
             want_final=True):
    """xproj [T,B,ndir*4H] fp32 (interleaved gate layout [T,B,ndir,H,4]).
    Returns gates, hout, cs, c_final, h_final."""
    h = _h(xproj)
    _chk(xproj, torch.float32, 'xproj')
    _chk(seq_len, torch.int32, 'seq_len')
    T, B, G = xproj.shape
    if G != ndir * 4 * H:
        raise ValueError('xproj last dim %d != ndir*4H' % G)
    dev = xproj.device
    gates = torch.empty((T, B, ndir * 4 * H), dtype=TORCH_DTYPE[dtype], device=dev)
    hout = torch.empty((T, B, ndir * H), dtype=TORCH_DTYPE[dtype], device=dev)
    cs = torch.empty((T, B, ndir * H), dtype=torch.float32, device=dev)
    cf = torch.empty((ndir, B, H), dtype=torch.float32, device=dev) if want_final else None
    hf = torch.empty((ndir, B, H), dtype=torch.float32, device=dev) if want_final else None
    h.check(h.lib.asr_lstm_fwd(h.h, dtype, T, B, H, ndir, _p(xproj), _p(wh_packed), _p(peep),
                               _p(seq_len), float(forget_bias), float(cell_clip or 0.0), _p(gates),
                               _p(hout), _p(cs), _p(cf), _p(hf), _s()), 'asr_lstm_fwd')
    return gates, hout, cs, cf, hf


def lstm_bwd(dhout, gates, cs, wh_packed_bwd, peep, seq_len, H, ndir, dtype, d_c_final=None,
             d_h_final=None, want_dpeep=True, clip_no_grad=0.0):
    """clip_no_grad > 0 (fp32 only): the forward's cell_clip, for a cell whose clamp passes no gradient (LSTMCell's
    tf.clip_by_value; LSTMBlockCell's clip is straight-through: 0)."""
    h = _h(dhout)
    _chk(dhout, torch.float32, 'dhout')
    T, B, _ = dhout.shape
    dev = dhout.device
    dgates = torch.empty((T, B, ndir * 4 * H), dtype=TORCH_DTYPE[dtype], device=dev)
    dpeep = ws = None
    if want_dpeep:
        dpeep = torch.empty((ndir, 7, H), dtype=torch.float32, device=dev)   # 3 peephole + 4 bias rows
        ws = torch.empty(((B // 16) * ndir * 7 * H,), dtype=torch.float32, device=dev)
    h.check(h.lib.asr_lstm_bwd_ex(h.h, dtype, T, B, H, ndir, _p(dhout), _p(gates), _p(cs),
                                  _p(wh_packed_bwd), _p(peep), _p(seq_len), _p(d_c_final), _p(d_h_final),
                                  float(clip_no_grad or 0.0), _p(dgates), _p(dpeep), _p(ws), _s()), 'asr_lstm_bwd')
    return dgates, dpeep


def gru_fwd(xg, xc, wgh, wch, seq_len, tmax, H, ndir):
    """xg [T,B,ndir*2H], xc [T,B,ndir*H] fp32 (hoisted x-projections + biases); wgh [ndir,H,2H], wch [ndir,H,H].
    Returns dict(r, u, c, rh [T,B,ndir*H] each, hout [T,B,ndir*H], h_final [ndir,B,H])."""
    h = _h(xg)
    _chk(xg, torch.float32, 'xg')
    _chk(seq_len, torch.int32, 'seq_len')
    T, B, G = xg.shape
    if G != ndir * 2 * H or tuple(xc.shape) != (T, B, ndir * H):
        raise ValueError('gru_fwd: xg / xc shapes do not match ndir, H')
    dev = xg.device
    out = {k: torch.empty((T, B, ndir * H), dtype=torch.float32, device=dev) for k in ('r', 'u', 'c', 'hout')}
    # rh is contracted over ALL T*B rows into the candidate kernel's recurrent gradient (with dcand = 0 at padded frames):
    # the kernel skips inactive rows, so they must hold zeros, not allocator leftovers (0 * NaN = NaN)
    out['rh'] = torch.zeros((T, B, ndir * H), dtype=torch.float32, device=dev)
    hs = torch.empty((2, ndir, B, H), dtype=torch.float32, device=dev)
    h.check(h.lib.asr_gru_fwd(h.h, T, B, H, ndir, _p(xg), _p(xc), _p(wgh), _p(wch), _p(seq_len), int(tmax),
                              _p(out['r']), _p(out['u']), _p(out['c']), _p(out['rh']), _p(out['hout']), _p(hs), _s()),
            'asr_gru_fwd')
    out['h_final'] = hs[0]
    return out


def gru_bwd(dout, d_h_final, saved, wghT, wchT, seq_len, tmax, H, ndir):
    """dout [T,B,ndir*H] fp32 -> (dgate [T,B,ndir*2H], dcand [T,B,ndir*H]) pre-activation gradients."""
    h = _h(dout)
    _chk(dout, torch.float32, 'dout')
    T, B, _ = dout.shape
    dev = dout.device
    dgate = torch.empty((T, B, ndir * 2 * H), dtype=torch.float32, device=dev)
    dcand = torch.empty((T, B, ndir * H), dtype=torch.float32, device=dev)
    work = torch.empty((2, ndir, B, H), dtype=torch.float32, device=dev)
    h.check(h.lib.asr_gru_bwd(h.h, T, B, H, ndir, _p(dout), _p(d_h_final), _p(saved['hout']), _p(saved['r']),
                              _p(saved['u']), _p(saved['c']), _p(wghT), _p(wchT), _p(seq_len), int(tmax), _p(dgate),
                              _p(dcand), _p(work), _s()), 'asr_gru_bwd')
    return dgate, dcand


def check_async_errors(device=0, flush_deferred=True):
    """Device sync + sticky error word of the multi-CU LSTM kernels (raises on a hand-off timeout).
    The blocking form: call at sync points (evaluation, checkpoint, end of an epoch)."""
    h = _lib.handle(device)
    flags = C.c_uint(0)
    rc = h.lib.asr_check_async_errors(h.h, C.byref(flags))
    dkey = device.index or 0 if isinstance(device, torch.device) else int(device)
    for (d, lane), h2 in list(_lib._handles.items()):
        # recurrences also run on the second pipeline's lane (models/encoders/core/blstm.py PIPE_LANE): its handle has
        # its own exchange areas and its own error word
        if d == dkey and lane == WATCHED_SIDE_LANE:
            f2 = C.c_uint(0)
            rc2 = h2.lib.asr_check_async_errors(h2.h, C.byref(f2))
            if f2.value:
                h2.lib.asr_clear_async_errors(h2.h, _s())
                flags.value |= f2.value
                rc = rc or rc2
                h = h2 if rc2 else h
    if flags.value:
        h0 = _lib.handle(device)
        h0.lib.asr_clear_async_errors(h0.h, _s())     # sticky until reported once
        # ... and once only: the non-blocking watch may still hold copies of the same word armed before this point (the
        # device has been drained, they have all landed); left in its ring they would raise the error a second time up to
        # DEPTH optimizer steps later -- after the caller has restored its checkpoint
        w = _watches.get(device.index or 0 if isinstance(device, torch.device) else int(device))
        if w is not None:
            w.events = [None] * w.DEPTH
    h.check(rc, 'asr_check_async_errors')
    if flush_deferred:
        _deferred_for(device).flush()                # the device is idle: every counter armed on it has landed
    return flags.value


WATCHED_SIDE_LANE = 6     # == blstm.PIPE_LANE: the one side lane recurrence kernels are issued on


class ErrorWatch(object):
    """Non-blocking watch on the sticky error word of the cluster recurrence kernels.

    poll() is called once per optimizer step (models/model_base.py Optimizer.apply_gradients): it enqueues a 4-byte
    device->host copy of the word behind the step's kernels and inspects the copy armed DEPTH polls earlier (its event
    has long completed, so the host never stalls in steady state).  A hand-off timeout therefore raises AsrError at
    most DEPTH steps after it happened instead of silently corrupting the weights."""
    DEPTH = 3

    def __init__(self, device):
        self.dev = device
        self.host = torch.zeros(2 * self.DEPTH, dtype=torch.int32).pin_memory()   # [slot] main handle, [DEPTH + slot] side lane's
        self.events = [None] * self.DEPTH
        self.i = 0
        self.waited_s = 0.0   # host time spent blocked on a step armed DEPTH polls ago (bench.py subtracts it)

    def poll(self):
        slot = self.i % self.DEPTH
        ev = self.events[slot]
        if ev is not None:
            if not ev.query():
                # the host is DEPTH steps ahead of the device: this wait is the throttle of the issue loop, not issue work
                t0 = _time.perf_counter()
                ev.synchronize()
                self.waited_s += _time.perf_counter() - t0
            flags = int(self.host[slot]) | int(self.host[self.DEPTH + slot])
            if flags:
                h = _lib.handle(self.dev)
                h.lib.asr_clear_async_errors(h.h, _s())
                h2 = _lib._handles.get((self.dev, WATCHED_SIDE_LANE))
                if h2 is not None:
                    h2.lib.asr_clear_async_errors(h2.h, _s())
                self.events = [None] * self.DEPTH
                if flags & 3:
                    raise _lib.AsrError('LSTM cluster hand-off timed out (flags 0x%x): the recurrent state of a recent '
                                        'step is garbage -- restore the last checkpoint' % flags)
                # not a hand-off fault: the MODEL diverged.  The reference would carry a NaN loss on (and its learning-rate
                # controller / early stop would react); the bf16 cluster exchange cannot represent a non-finite h (its tag
                # bits would launder it into a finite value for the peers), so this path stops instead (INTEGRATION.md,
                # "Deviations"; bf16 self-tagged exchange only -- the fp32 paths propagate NaN like the reference)
                raise _lib.AsrError('LSTM recurrence produced a non-finite hidden state (flags 0x%x): the model diverged '
                                    '(learning rate / clipping); weights of the last %d steps are NaN-contaminated'
                                    % (flags, self.DEPTH))
        h = _lib.handle(self.dev)
        h.check(h.lib.asr_peek_async_errors(h.h, C.c_void_p(self.host.data_ptr() + 4 * slot), _s()),
                'asr_peek_async_errors')
        self.host[self.DEPTH + slot] = 0
        h2 = _lib._handles.get((self.dev, WATCHED_SIDE_LANE))
        if h2 is not None:       # the second pipeline's handle (its work has been joined into this stream by now)
            h2.check(h2.lib.asr_peek_async_errors(h2.h, C.c_void_p(self.host.data_ptr() + 4 * (self.DEPTH + slot)), _s()),
                     'asr_peek_async_errors')
        ev = torch.cuda.Event()
        ev.record(_cur_stream())
        self.events[slot] = ev
        self.i += 1


_watches = {}


class DeferredCheck(object):
    """A device-side counter that must be zero (tf.nn.ctc_loss's "Not enough time for target transition sequence"
    InvalidArgumentError, ctc.py:289 with ignore_longer_outputs_than_inputs=False), checked WITHOUT stalling the training
    step: arm() copies the counter to pinned memory behind the step's kernels -- on the counter's OWN device and that
    device's current stream (one DeferredCheck per device) -- and every earlier copy that has landed is inspected then.
    Lateness: the reference raises inside the sess.run of the offending step; here the error surfaces at a later arm() --
    in steady state within ErrorWatch.DEPTH (3) optimizer steps, because that is how far the issue loop may run ahead of
    the device, and NEVER more than DEPTH (4) optimizer STEPS late, however many CTC heads arm a counter per step
    (MultitaskCTC arms two): a copy armed DEPTH steps ago is waited for (the device's ErrorWatch counts the steps:
    note_step()); without optimizer steps in between (a loop of compute_loss calls) at most RING - 1 copies are pending.
    The optimizer has applied the updates of the steps in between.  An error DROPS every other pending copy (another
    head's, the later steps' -- which ran on the state the failing step left and would only repeat the report while the
    caller restores its checkpoint): one exception per incident.  flush() is the blocking form; it runs at every sync point: evaluation (is_training=False), Saver.save / check_async_errors (checkpoints), the
    end of the recipes' epochs, and at interpreter exit (a pending error is printed, it cannot be raised any more)."""
    DEPTH = 4          # optimizer steps
    RING = 16          # pinned slots: >= DEPTH x the heads of any model here, and the cap when no optimizer steps run

    def __init__(self, device=None):
        self.device = device
        self.slots = []          # (pinned host tensor, event, exception factory, step it was armed in)
        self.waited_s = 0.0
        self.ring = None
        self.n = 0
        self.step = 0

    def note_step(self):
        """One optimizer step has been issued on this device (watch_async_errors)."""
        self.step += 1

    def _inspect(self, host, ev, make_exc, step, block):
        if not ev.query():
            if not block:
                return False
            t0 = _time.perf_counter()
            ev.synchronize()
            self.waited_s += _time.perf_counter() - t0
        n = int(host[0])
        if n:
            self.slots = []
            # a timed-out cluster hand-off leaves NaN activations, which the CTC kernels count as infeasible rows: report the
            # root cause (AsrError from the sticky error word) rather than its symptom
            check_async_errors(self.device if self.device is not None else torch.cuda.current_device(), flush_deferred=False)
            raise make_exc(n)
        return True

    def arm(self, counter, make_exc):
        if self.ring is None:        # one pinned block for the life of the process: no pinned allocation per step
            self.ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
        self.n += 1
        host = self.ring[self.n % self.RING:][:1]
        host.copy_(counter.view(-1)[:1].to(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(_cur_stream(counter.device))       # the stream the copy was enqueued on: the counter's device
        self.slots.append((host, ev, make_exc, self.step))
        while self.slots:
            block = self.step - self.slots[0][3] >= self.DEPTH or len(self.slots) >= self.RING - 1
            if not self._inspect(*self.slots[0], block=block):
                break
            self.slots.pop(0)

    def flush(self):
        while self.slots:
            slot = self.slots.pop(0)
            self._inspect(*slot, block=True)


_deferred_by_dev = {}


def _deferred_for(device):
    d = device.index if isinstance(device, torch.device) else device
    d = torch.cuda.current_device() if d is None else int(d)
    w = _deferred_by_dev.get(d)
    if w is None:
        w = _deferred_by_dev[d] = DeferredCheck(d)
    return w


def defer_zero_check(counter, make_exc, blocking=False):
    """counter: device int tensor that must be 0; make_exc(n) builds the exception.  blocking=True checks now."""
    if not counter.is_cuda:
        if int(counter.view(-1)[0]):
            raise make_exc(int(counter.view(-1)[0]))
        return
    w = _deferred_for(counter.device)
    with torch.cuda.device(counter.device):
        w.arm(counter, make_exc)
        if blocking:
            w.flush()


def flush_deferred_checks():
    """Blocking inspection of every armed counter on every device (end of an epoch / of training)."""
    for w in list(_deferred_by_dev.values()):
        w.flush()


def _flush_deferred_at_exit():
    try:
        flush_deferred_checks()
    except Exception as e:       # too late to raise into the training loop: say so
        import sys
        sys.stderr.write('tensorflow_end2end_speech_recognition_amd: an error of one of the LAST training steps was still '
                         'pending at exit: %r\n' % (e,))


import atexit as _atexit      # noqa: E402
_atexit.register(_flush_deferred_at_exit)


def watch_waited_seconds(device=0):
    """Cumulative host time the device's ErrorWatch spent blocked behind the GPU (0 if none exists yet)."""
    w = _watches.get(device.index or 0 if isinstance(device, torch.device) else int(device))
    return (w.waited_s if w is not None else 0.0) + sum(d.waited_s for d in _deferred_by_dev.values())


def watch_async_errors(device):
    """One ErrorWatch per device; see ErrorWatch.poll."""
    dev = device.index or 0 if isinstance(device, torch.device) else int(device)
    w = _watches.get(dev)
    if w is None:
        w = _watches[dev] = ErrorWatch(dev)
    d = _deferred_by_dev.get(dev)
    if d is not None:
        d.note_step()
    w.poll()


def debug_set_lstm_flags(flags):
    """ASR_LSTM_DFLAGS at run time (tests): 16 = write-through exchange, 64 = force hand-off timeouts."""
    _lib.load().asr_debug_set_lstm_flags(int(flags))


# ---------------------------------------------------------------- CTC
def debug_set_gru_persistent(on):
    """asr_debug_set_gru_persistent: 1 = one persistent launch per GRU layer call (default), 0 = launch per step."""
    _lib.load().asr_debug_set_gru_persistent(int(bool(on)))


def ctc_loss(logits, labels_flat, label_offsets, seq_len, max_label_len, grad_scale=1.0,
             want_grad=True):
    """logits [T,B,C] fp32; returns (loss [B], grad [T,B,C] or None, num_infeasible [1] int32)."""
    h = _h(logits)
    _chk(logits, torch.float32, 'logits')
    _chk(labels_flat, torch.int32, 'labels_flat')
    _chk(label_offsets, torch.int32, 'label_offsets')
    _chk(seq_len, torch.int32, 'seq_len')
    T, B, Cc = logits.shape
    dev = logits.device
    nbytes = h.lib.asr_ctc_workspace_bytes(T, B, int(max_label_len))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    grad = torch.empty_like(logits) if want_grad else None
    ninf = torch.empty((1,), dtype=torch.int32, device=dev)     # zeroed by the call
    h.check(h.lib.asr_ctc_loss(h.h, _p(logits), T, B, Cc, _p(labels_flat), _p(label_offsets),
                               _p(seq_len), int(max_label_len), float(grad_scale), _p(loss), _p(grad),
                               _p(ninf), _p(ws), nbytes, _s()), 'asr_ctc_loss')
    return loss, grad, ninf


def ctc_greedy_decode(logits, seq_len, blank=None):
    h = _h(logits)
    _chk(logits, torch.float32, 'logits')
    _chk(seq_len, torch.int32, 'seq_len')
    T, B, Cc = logits.shape
    if blank is None:
        blank = Cc - 1
    out = torch.empty((B, T), dtype=torch.int32, device=logits.device)
    n = torch.empty((B,), dtype=torch.int32, device=logits.device)
    h.check(h.lib.asr_ctc_greedy_decode(h.h, _p(logits), T, B, Cc, _p(seq_len), int(blank), _p(out),
                                        _p(n), _s()), 'asr_ctc_greedy_decode')
    return out, n


def ctc_beam_decode(logits, seq_len, beam_width, blank=None):
    """Prefix beam search.  logits [T,B,C] fp32 -> (labels [B,T] int32 padded -1, lengths [B],
    scores [B] float64 = -log p of the best prefix)."""
    h = _h(logits)
    _chk(logits, torch.float32, 'logits')
    _chk(seq_len, torch.int32, 'seq_len')
    T, B, Cc = logits.shape
    if blank is None:
        blank = Cc - 1
    dev = logits.device
    nbytes = h.lib.asr_ctc_beam_workspace_bytes(T, B, Cc, int(beam_width))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    out = torch.empty((B, T), dtype=torch.int32, device=dev)
    n = torch.empty((B,), dtype=torch.int32, device=dev)
    score = torch.empty((B,), dtype=torch.float64, device=dev)
    h.check(h.lib.asr_ctc_beam_decode(h.h, _p(logits), T, B, Cc, _p(seq_len), int(blank), int(beam_width),
                                      _p(out), _p(n), _p(score), _p(ws), nbytes, _s()), 'asr_ctc_beam_decode')
    return out, n, score


def softmax_rows(x2d):
    h = _h(x2d)
    _chk(x2d, torch.float32, 'x')
    out = torch.empty_like(x2d)
    h.check(h.lib.asr_softmax_rows(h.h, _p(x2d), _p(out), x2d.shape[0], x2d.shape[1], _s()),
            'asr_softmax_rows')
    return out


# ---------------------------------------------------------------- attention decoder
def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _col_block(t, B, W, name):
    """(pointer, row stride) of a [B, W] column block of a row-major array (unit inner stride)."""
    if t is None:
        return None, 0
    if t.dim() != 2 or tuple(t.shape) != (B, W) or t.stride(1) != 1 or t.dtype != torch.float32:
        raise ValueError('%s must be an fp32 [B,%d] column block with unit inner stride' % (name, W))
    return C.c_void_p(t.data_ptr()), t.stride(0)


def lstm_cell_fwd(pre, c_prev, h_prev, peep, live, forget_bias=1.0, cell_clip=0.0, out_mask=None, want_cell_out=False,
                  h_also=None, cell_out_also=None):
    """One decoder cell step.  Returns (gates, c_raw, c_out, h_out, h_raw) -- plus cell_out = h_raw * out_mask as a
    sixth element when want_cell_out.  h_also / cell_out_also: [B,U] column blocks of wider arrays (the next step's
    cell input, the attentional vector's input) that receive h_out / cell_out as well, inside the same launch."""
    h = _h(pre)
    B, U4 = pre.shape
    U = U4 // 4
    dev = pre.device
    gates, c_raw, c_out, h_out, h_raw = _f32((B, U4), dev), _f32((B, U), dev), _f32((B, U), dev), \
        _f32((B, U), dev), _f32((B, U), dev)
    cell_out = _f32((B, U), dev) if want_cell_out else None
    hp, hld = _col_block(h_also, B, U, 'h_also')
    cp, cld = _col_block(cell_out_also, B, U, 'cell_out_also')
    h.check(h.lib.asr_lstm_cell_fwd_ex(h.h, _p(pre), _p(c_prev), _p(h_prev), _p(peep), _p(live), B, U,
                                       float(forget_bias), float(cell_clip or 0.0), _p(gates), _p(c_raw), _p(c_out),
                                       _p(h_out), _p(h_raw), _p(out_mask), _p(cell_out), hp, hld, cp, cld, _s()),
            'asr_lstm_cell_fwd')
    if want_cell_out:
        return gates, c_raw, c_out, h_out, h_raw, cell_out
    return gates, c_raw, c_out, h_out, h_raw


def lstm_cell_gemm_prep(W, bias):
    """Gate-interleaved image [(K + 1), 4U] of a decoder cell's kernel [K,4U] and bias [4U] (asr_lstm_cell_gemm_prep)."""
    h = _h(W)
    K, U4 = W.shape
    out = _f32(((K + (1 if bias is not None else 0)), U4), W.device)
    h.check(h.lib.asr_lstm_cell_gemm_prep(h.h, _p(W), _p(bias), K, U4 // 4, _p(out), _s()), 'asr_lstm_cell_gemm_prep')
    return out


def lstm_cell_gemm_fwd(x, W_il, has_bias, c_prev, h_prev, peep, live, forget_bias=1.0, cell_clip=0.0, out_mask=None,
                       h_also=None, cell_out_also=None):
    """lstm_cell_fwd(gemm(x, W) + b, ...) as one launch on the interleaved image; x [B,K] (a row block of a wider array is
    fine).  Returns (gates, c_raw, c_out, h_out, h_raw, cell_out)."""
    h = _h(x)
    B, K = x.shape
    U = W_il.shape[1] // 4
    dev = x.device
    gates, c_raw, c_out, h_out, h_raw, cell_out = _f32((B, 4 * U), dev), _f32((B, U), dev), _f32((B, U), dev), \
        _f32((B, U), dev), _f32((B, U), dev), _f32((B, U), dev)
    hp, hld = _col_block(h_also, B, U, 'h_also')
    cp, cld = _col_block(cell_out_also, B, U, 'cell_out_also')
    if x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError('lstm_cell_gemm_fwd: x must be fp32 with unit inner stride')
    h.check(h.lib.asr_lstm_cell_gemm_fwd(h.h, C.c_void_p(x.data_ptr()), int(x.stride(0)), K, _p(W_il), int(bool(has_bias)), _p(c_prev), _p(h_prev),
                                         _p(peep), _p(live), B, U, float(forget_bias), float(cell_clip or 0.0), _p(gates),
                                         _p(c_raw), _p(c_out), _p(h_out), _p(h_raw), _p(out_mask), _p(cell_out), hp, hld, cp,
                                         cld, _s()), 'asr_lstm_cell_gemm_fwd')
    return gates, c_raw, c_out, h_out, h_raw, cell_out


def lstm_cell_gemm_prep_h(W, bias):
    """The bf16 images of a decoder cell's kernel [K,4U] (+ bias) for lstm_cell_gemm_fwd_h / _bwd_h: one opaque buffer."""
    h = _h(W)
    K, U4 = W.shape
    nbytes = int(h.lib.asr_lstm_cell_gemm_h_bytes(K, U4 // 4))
    img = torch.empty((nbytes // 4,), dtype=torch.float32, device=W.device)
    h.check(h.lib.asr_lstm_cell_gemm_prep_h(h.h, _p(W), _p(bias), K, U4 // 4, _p(img), _s()), 'asr_lstm_cell_gemm_prep_h')
    return img


def lstm_cell_gemm_fwd_h(x, img, U, c_prev, h_prev, peep, live, forget_bias=1.0, cell_clip=0.0, out_mask=None,
                         h_also=None, cell_out_also=None):
    """lstm_cell_gemm_fwd with bf16 weights (the image of lstm_cell_gemm_prep_h)."""
    h = _h(x)
    B, K = x.shape
    dev = x.device
    if x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError('lstm_cell_gemm_fwd_h: x must be fp32 with unit inner stride')
    gates, c_raw, c_out, h_out, h_raw, cell_out = _f32((B, 4 * U), dev), _f32((B, U), dev), _f32((B, U), dev), \
        _f32((B, U), dev), _f32((B, U), dev), _f32((B, U), dev)
    hp, hld = _col_block(h_also, B, U, 'h_also')
    cp, cld = _col_block(cell_out_also, B, U, 'cell_out_also')
    h.check(h.lib.asr_lstm_cell_gemm_fwd_h(h.h, C.c_void_p(x.data_ptr()), int(x.stride(0)), K, _p(img), _p(c_prev), _p(h_prev),
                                           _p(peep), _p(live), B, U, float(forget_bias), float(cell_clip or 0.0), _p(gates),
                                           _p(c_raw), _p(c_out), _p(h_out), _p(h_raw), _p(out_mask), _p(cell_out), hp, hld,
                                           cp, cld, _s()), 'asr_lstm_cell_gemm_fwd_h')
    return gates, c_raw, c_out, h_out, h_raw, cell_out


def lstm_cell_gemm_bwd_h(dpre, img, K):
    """dx [B,K] = dpre [B,4U] W^T on the bf16 copy of W inside the image."""
    h = _h(dpre)
    B, U4 = dpre.shape
    dx = _f32((B, K), dpre.device)
    h.check(h.lib.asr_lstm_cell_gemm_bwd_h(h.h, _p(dpre), B, K, U4 // 4, _p(img), _p(dx), K, _s()), 'asr_lstm_cell_gemm_bwd_h')
    return dx


def lstm_cell_bwd(dh_use, dc_next, dh_next, gates, c_raw, c_prev, peep, live, want_dpeep=True, dpre_out=None,
                  dpeep_out=None, cell_clip=0.0):
    """dpre_out [B,4U] / dpeep_out [B,3U] (contiguous rows of the caller's per-step arrays): written in place.
    cell_clip: the clip the forward applied to the new cell state (0: none) -- a clamped state passes no gradient."""
    h = _h(dh_use)
    B, U = dh_use.shape
    dev = dh_use.device
    dpre = dpre_out if dpre_out is not None else _f32((B, 4 * U), dev)
    dc_prev, dh_carry = _f32((B, U), dev), _f32((B, U), dev)
    dpeep = (dpeep_out if dpeep_out is not None else _f32((B, 3, U), dev)) if want_dpeep else None
    h.check(h.lib.asr_lstm_cell_bwd_ex(h.h, _p(dh_use), _p(dc_next), _p(dh_next), _p(gates), _p(c_raw), _p(c_prev),
                                       _p(peep), _p(live), B, U, float(cell_clip or 0.0), _p(dpre), _p(dc_prev),
                                       _p(dh_carry), _p(dpeep), _s()), 'asr_lstm_cell_bwd')
    return dpre, dc_prev, dh_carry, dpeep


def att_energy_fwd(keys, qz, v, T, mode):
    h = _h(qz)
    B, A = qz.shape
    energy = _f32((B, T), qz.device)
    h.check(h.lib.asr_att_energy_fwd(h.h, _p(keys), _p(qz), _p(v), T, B, A, int(mode), _p(energy), _s()),
            'asr_att_energy_fwd')
    return energy


def att_energy_bwd(denergy, keys, qz, v, mode, dkeys=None, want_dv=True, dqz_out=None, dv_out=None):
    h = _h(qz)
    B, A = qz.shape
    T = denergy.shape[1]
    dqz = dqz_out if dqz_out is not None else _f32((B, A), qz.device)
    dv = (dv_out if dv_out is not None else _f32((B, A), qz.device)) if want_dv else None
    h.check(h.lib.asr_att_energy_bwd(h.h, _p(denergy), _p(keys), _p(qz), _p(v), T, B, A, int(mode), _p(dkeys),
                                     _p(dqz), _p(dv), _s()), 'asr_att_energy_bwd')
    return dqz, dv


def att_softmax_ctx_fwd(energy, seq_len, sharpening, enc, alpha_out=None, sigmoid_norm=None, ctx_also=()):
    """sigmoid_norm: None = softmax; a [B] fp32 tensor = sigmoid smoothing (receives sum_t sigmoid(e)).
    ctx_also: up to two [B,E] column blocks of wider arrays that receive the context as well (same launch)."""
    h = _h(energy)
    B, T = energy.shape
    E = enc.shape[2]
    alpha = alpha_out if alpha_out is not None else _f32((B, T), energy.device)
    ctx = _f32((B, E), energy.device)
    if sigmoid_norm is not None and (sigmoid_norm.dtype != torch.float32 or sigmoid_norm.numel() != B):
        raise ValueError('sigmoid_norm must be fp32 [B]')
    also = [t for t in ctx_also if t is not None]
    if len(also) > 2:
        raise ValueError('att_softmax_ctx_fwd: at most two extra destinations')
    p2, l2 = _col_block(also[0] if len(also) > 0 else None, B, E, 'ctx_also')
    p3, l3 = _col_block(also[1] if len(also) > 1 else None, B, E, 'ctx_also')
    h.check(h.lib.asr_att_softmax_ctx_fwd_ex(h.h, _p(energy), _p(seq_len), float(sharpening), _p(enc),
                                             dtype_id(enc.dtype), T, B, E, _p(alpha), _p(ctx), _p(sigmoid_norm),
                                             p2, l2, p3, l3, _s()), 'asr_att_softmax_ctx_fwd')
    return alpha, ctx


def att_loc_energy_fwd(alpha_prev, filt, wfil, keys, qz, v, T):
    """Location / hybrid energies with the previous step's weights carried through conv1d -> W_filter
    (attention_layer.py:191-265).  alpha_prev [B,T]; filt [taps,1,10]; wfil [10,A]; keys [T,B,A] or None."""
    h = _h(qz)
    B, A = qz.shape
    taps = filt.shape[0]
    energy = _f32((B, T), qz.device)
    h.check(h.lib.asr_att_loc_energy_fwd(h.h, _p(alpha_prev), _p(filt), _p(wfil), _p(keys), _p(qz), _p(v), T, B, A,
                                         taps, _p(energy), _s()), 'asr_att_loc_energy_fwd')
    return energy


def att_loc_energy_bwd(denergy, alpha_prev, filt, wfil, keys, qz, v, dwfil_rows, dfilt_rows, accumulate, dkeys=None,
                       dqz_out=None, dv_out=None):
    """-> (dqz [B,A], dv_rows [B,A], dalpha_prev [B,T]); dwfil_rows [B,10,A] / dfilt_rows [B,taps,10] are
    overwritten (accumulate False) or added to (True); dkeys += in place when given."""
    h = _h(qz)
    B, A = qz.shape
    T = denergy.shape[1]
    taps = filt.shape[0]
    dqz = dqz_out if dqz_out is not None else _f32((B, A), qz.device)
    dv = dv_out if dv_out is not None else _f32((B, A), qz.device)
    dap = _f32((B, T), qz.device)
    h.check(h.lib.asr_att_loc_energy_bwd(h.h, _p(denergy), _p(alpha_prev), _p(filt), _p(wfil), _p(keys), _p(qz), _p(v),
                                         T, B, A, taps, _p(dkeys), _p(dqz), _p(dv), _p(dwfil_rows), _p(dfilt_rows),
                                         _p(dap), 1 if accumulate else 0, _s()), 'asr_att_loc_energy_bwd')
    return dqz, dv, dap


def att_softmax_ctx_bwd(dctx, alpha, seq_len, sharpening, enc, denc=None, sigmoid_norm=None, dalpha_extra=None):
    """denc None: only denergy is produced; the caller accumulates d_enc = sum_steps alpha (x) dctx itself.
    sigmoid_norm: the tensor the forward filled (sigmoid smoothing), or None (softmax).
    dalpha_extra [B,T]: gradient w.r.t. alpha from the next step's carried location features, or None."""
    h = _h(dctx)
    B, T = alpha.shape
    E = enc.shape[2]
    denergy = _f32((B, T), dctx.device)
    if sigmoid_norm is not None and (sigmoid_norm.dtype != torch.float32 or sigmoid_norm.numel() != B):
        raise ValueError('sigmoid_norm must be fp32 [B]')
    h.check(h.lib.asr_att_softmax_ctx_bwd(h.h, _p(dctx), _p(alpha), _p(seq_len), float(sharpening), _p(enc),
                                          dtype_id(enc.dtype), T, B, E, _p(denergy), _p(denc), _p(sigmoid_norm),
                                          _p(dalpha_extra), _s()),
            'asr_att_softmax_ctx_bwd')
    return denergy


class _AttDecoder(C.Structure):
    """struct asr_att_decoder (include/asr_hip.h), field for field."""
    _INTS = ['To', 'B', 'T', 'U', 'Em', 'E2', 'A', 'att_mode', 'has_query_fc', 'carry_alpha', 'taps', 'enc_dtype']
    _FLOATS = ['forget_bias', 'cell_clip', 'sharpening']
    _PTRS1 = ['W_cell', 'b_cell', 'peep', 'W_q', 'b_q', 'v']
    _PTRS2 = ['keys', 'enc', 'seq_len', 'filt', 'wfil', 'alpha_zero', 'live', 'dmask', 'dec_in', 'av_in', 'alpha_all',
              'snorm_all', 'gates_all', 'craw_all', 'c_all', 'h_all', 'qz_all', 'work', 'dav_cell', 'dav_ctx', 'dctx_all',
              'dpre_all', 'dqz_all', 'dv_all', 'dpeep_all', 'd_in_all', 'dkeys', 'dwfil_rows', 'dfilt_rows', 'dc0', 'dh0',
              'W_cell_il', 'W_cell_h']
    _fields_ = ([(n, C.c_int) for n in _INTS] + [(n, C.c_float) for n in _FLOATS] +
                [(n, C.c_void_p) for n in _PTRS1] + [('ld_wq', C.c_int)] + [(n, C.c_void_p) for n in _PTRS2])


def _att_decoder_struct(a):
    st = _AttDecoder()
    for n in _AttDecoder._INTS:
        setattr(st, n, int(a.get(n, 0) or 0))
    for n in _AttDecoder._FLOATS:
        setattr(st, n, float(a.get(n, 0.0) or 0.0))
    for n in _AttDecoder._PTRS1 + _AttDecoder._PTRS2:
        t = a.get(n)
        if t is not None and not t.is_cuda:
            raise RuntimeError('HIP path needs a CUDA(ROCm) tensor; there is no CPU fallback')
        if t is not None and n != 'W_q' and not t.is_contiguous():
            raise ValueError('att_decoder: %s must be contiguous' % n)
        setattr(st, n, t.data_ptr() if t is not None else None)
    wq = a.get('W_q')
    st.ld_wq = int(wq.stride(0)) if wq is not None else 0
    return st


FUSED_CELL_GEMM = os.environ.get('ASR_DEC_CELL_GEMM', '1') != '0'   # A/B: 0 keeps product and cell as two launches


BF16_CELL_WEIGHTS = os.environ.get('ASR_DEC_CELL_BF16', '1') != '0'   # A/B: 0 keeps fp32 decoder weights in bf16 models


def _cell_gemm_image(h, a):
    """Work space for the gate-interleaved image of W_cell | b_cell (the loops fill it): with it a decoder step's cell-input
    product and LSTM cell are one launch (asr_lstm_cell_gemm_fwd).  a['cell_bf16'] (bf16-operand models): the bf16 images
    instead (asr_lstm_cell_gemm_*_h: forward product + cell, backward product)."""
    Din = a['Em'] + a['E2'] + a['U']
    if not (FUSED_CELL_GEMM and h.lib.asr_lstm_cell_gemm_ok(int(a['B']), Din, int(a['U']), Din)):
        return
    if a.get('cell_bf16') and BF16_CELL_WEIGHTS and a['U'] % 16 == 0:
        if a.get('W_cell_h') is None:
            nbytes = int(h.lib.asr_lstm_cell_gemm_h_bytes(Din, int(a['U'])))
            a['W_cell_h'] = torch.empty((nbytes // 4,), dtype=torch.float32, device=a['dec_in'].device)
    elif a.get('W_cell_il') is None:
        a['W_cell_il'] = _f32(((Din + 1) * 4 * a['U'],), a['dec_in'].device)


def att_decoder_fwd(a):
    """All To steps of the attention decoder's forward pass from one call (asr_att_decoder_fwd).  `a`: dict of the
    struct's fields (ints / floats / cuda tensors or None); the per-step arrays are filled in place."""
    h = _h(a['dec_in'])
    if a.get('work') is None:
        a['work'] = _f32((a['B'] * (5 * a['U'] + a['T'] + a['E2']),), a['dec_in'].device)
    _cell_gemm_image(h, a)
    st = _att_decoder_struct(a)
    h.check(h.lib.asr_att_decoder_fwd(h.h, C.byref(st), _s()), 'asr_att_decoder_fwd')


def att_decoder_bwd(a):
    """The reverse loop (asr_att_decoder_bwd): fills dctx_all, dpre_all, dqz_all, dv_all, dpeep_all, d_in_all, dc0, dh0
    (and adds into dkeys / the filter gradients)."""
    h = _h(a['dec_in'])
    a['work'] = _f32((a['B'] * (5 * a['U'] + 3 * a['T'] + a['E2']),), a['dec_in'].device)
    _cell_gemm_image(h, a)
    st = _att_decoder_struct(a)
    h.check(h.lib.asr_att_decoder_bwd(h.h, C.byref(st), _s()), 'asr_att_decoder_bwd')


class _AttInfer(C.Structure):
    """struct asr_att_infer (include/asr_hip.h), field for field."""
    _fields_ = [('W_av', C.c_void_p), ('W_out', C.c_void_p), ('b_out', C.c_void_p), ('embedding', C.c_void_p),
                ('C2', C.c_int), ('eos', C.c_int), ('live', C.c_void_p), ('av_all', C.c_void_p), ('logits_all', C.c_void_p),
                ('ids_all', C.c_void_p), ('live_count', C.c_void_p), ('host_live_count', C.c_void_p), ('check_every', C.c_int)]


def att_decoder_infer(a, W_av, W_out, b_out, embedding, eos, n_live, check_every=8):
    """Greedy inference loop (asr_att_decoder_infer): `a` as for att_decoder_fwd with To = max_decode_length and
    a['live'] a [To+1,B] tensor whose row 0 marks the rows that decode.  Returns a dict: ids [To,B] int32 (imputed),
    logits [To,B,C2], av [To,B,U], live [To+1,B], live_count [To+1] int32 (all on the device, nothing synchronised) and
    steps_issued (host int)."""
    h = _h(a['dec_in'])
    dev = a['dec_in'].device
    To, B, U = a['To'], a['B'], a['U']
    C2 = W_out.shape[1]
    if a.get('work') is None:
        a['work'] = _f32((B * (5 * U + a['T'] + a['E2']),), dev)
    _cell_gemm_image(h, a)
    out = dict(ids=torch.empty((To, B), dtype=torch.int32, device=dev), logits=_f32((To, B, C2), dev), av=_f32((To, B, U), dev),
               live=a['live'], live_count=torch.empty((To + 1,), dtype=torch.int32, device=dev))
    out['live_count'][:1].fill_(int(n_live))
    host = torch.ones((To + 1,), dtype=torch.int32).pin_memory() if check_every else None
    st = _att_decoder_struct(a)
    f = _AttInfer()
    for n, t in (('W_av', W_av), ('W_out', W_out), ('b_out', b_out), ('embedding', embedding), ('live', a['live']),
                 ('av_all', out['av']), ('logits_all', out['logits']), ('ids_all', out['ids']), ('live_count', out['live_count'])):
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError('att_decoder_infer: %s must be a contiguous device tensor' % n)
        setattr(f, n, t.data_ptr() if t is not None else None)
    f.C2, f.eos, f.check_every = int(C2), int(eos), int(check_every or 0)
    f.host_live_count = host.data_ptr() if host is not None else None
    issued = C.c_int(0)
    h.check(h.lib.asr_att_decoder_infer(h.h, C.byref(st), C.byref(f), C.byref(issued), _s()), 'asr_att_decoder_infer')
    out['steps_issued'] = issued.value
    out['_host'] = host                      # keeps the pinned words alive until the caller has synchronised
    return out


def tanh_fwd(x):
    h = _h(x)
    y = torch.empty_like(x)
    h.check(h.lib.asr_tanh_fwd(h.h, _p(x), _p(y), x.numel(), _s()), 'asr_tanh_fwd')
    return y


def tanh_bwd(dy, y):
    h = _h(dy)
    dx = torch.empty_like(dy)
    h.check(h.lib.asr_tanh_bwd(h.h, _p(dy), _p(y), _p(dx), dy.numel(), _s()), 'asr_tanh_bwd')
    return dx


def embedding_gather(W, ids):
    h = _h(W)
    _chk(ids, torch.int32, 'ids')
    out = _f32(tuple(ids.shape) + (W.shape[1],), W.device)
    h.check(h.lib.asr_embedding_gather(h.h, _p(W), _p(ids), ids.numel(), W.shape[1], _p(out), _s()),
            'asr_embedding_gather')
    return out


def embedding_scatter(dout, ids, vocab, out):
    h = _h(dout)
    E = dout.shape[-1]
    h.check(h.lib.asr_embedding_scatter(h.h, _p(dout), _p(ids), ids.numel(), E, int(vocab), _p(out), _s()),
            'asr_embedding_scatter')
    return out


def seq_xent(logits2d, targets, weights, eps, dscale, want_grad=True):
    h = _h(logits2d)
    rows, Cc = logits2d.shape
    row_loss = _f32((rows,), logits2d.device)
    dl = torch.empty_like(logits2d) if want_grad else None
    h.check(h.lib.asr_seq_xent(h.h, _p(logits2d), _p(targets), _p(weights), rows, Cc, float(eps), float(dscale),
                               _p(row_loss), _p(dl), _s()), 'asr_seq_xent')
    return row_loss, dl


def argmax_rows(x2d):
    h = _h(x2d)
    out = torch.empty((x2d.shape[0],), dtype=torch.int32, device=x2d.device)
    h.check(h.lib.asr_argmax_rows(h.h, _p(x2d), x2d.shape[0], x2d.shape[1], _p(out), _s()), 'asr_argmax_rows')
    return out


# ---------------------------------------------------------------- clip / decay / optimizers
class ClipPlan(object):
    """Device-side description of a flat fp32 parameter buffer split into tensors."""

    NORM_CHUNK = 4096   # == NORM_CHUNK in csrc/elementwise.hip (asr_clip_plan computes the same table)

    def __init__(self, offsets_host, device):
        off = np.asarray(offsets_host, dtype=np.int64)
        self.num_tensors = len(off) - 1
        cs = np.zeros(self.num_tensors + 1, dtype=np.int64)
        device = torch.device(device)
        if device.type == 'cuda':
            lib = _lib.load()
            hd = _lib.handle(device.index or 0)
            hd.check(lib.asr_clip_plan(hd.h, off.ctypes.data_as(C.c_void_p), self.num_tensors,
                                       cs.ctypes.data_as(C.c_void_p)), 'asr_clip_plan')
        else:   # host-only bookkeeping (CPU tests of the data-parallel logic); kernels still need a GPU
            cs[1:] = np.cumsum((np.diff(off) + self.NORM_CHUNK - 1) // self.NORM_CHUNK)
        self.total_chunks = int(cs[-1])
        self.offsets = torch.from_numpy(off).to(device)
        self.chunk_start = torch.from_numpy(cs).to(device)
        self.partial = torch.empty((max(self.total_chunks, 1),), dtype=torch.float32, device=device)


def clip_by_norm_multi(flat_grads, plan, clip_norm):
    h = _h(flat_grads)
    _chk(flat_grads, torch.float32, 'grads')
    h.check(h.lib.asr_clip_by_norm_multi(h.h, _p(flat_grads), _p(plan.offsets), _p(plan.chunk_start),
                                         plan.num_tensors, plan.total_chunks, float(clip_norm),
                                         _p(plan.partial), _s()), 'asr_clip_by_norm_multi')


def weight_decay(flat_grads, flat_params, plan, decay_mask, wd, l2_out=None):
    h = _h(flat_params)
    h.check(h.lib.asr_weight_decay(h.h, _p(flat_grads), _p(flat_params), _p(plan.offsets),
                                   _p(decay_mask), plan.num_tensors, float(wd), _p(l2_out), _s()),
            'asr_weight_decay')


def optimizer_step(opt_id, params, grads, slot0, slot1, lr, step):
    h = _h(params)
    h.check(h.lib.asr_optimizer_step(h.h, int(opt_id), _p(params), _p(grads), _p(slot0), _p(slot1),
                                     params.numel(), float(lr), int(step), _s()), 'asr_optimizer_step')


def scale_(x, s):
    h = _h(x)
    _chk(x, torch.float32, 'x')
    h.check(h.lib.asr_scale(h.h, _p(x), x.numel(), float(s), _s()), 'asr_scale')
    return x


# ---------------------------------------------------------------- data-parallel collective (RCCL through the C ABI)
class NativeComm(object):
    """asr_comm_* : one RCCL communicator per process / GPU, created from a 128-byte unique id that rank 0 generates
    (NativeComm.unique_id()) and the host program distributes.  allreduce_mean works in place on a flat fp32 cuda
    tensor, asynchronously on the current stream."""

    @staticmethod
    def _bind_library():
        import os
        lib = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        if os.path.exists(path):        # share the instance PyTorch has loaded
            lib.asr_comm_set_library(path.encode())
        return lib

    @staticmethod
    def unique_id():
        lib = NativeComm._bind_library()
        buf = C.create_string_buffer(128)
        rc = lib.asr_comm_unique_id(buf)
        if rc != 0:
            raise _lib.AsrError('asr_comm_unique_id failed (%d)' % rc)
        return bytes(buf.raw)

    def __init__(self, device, rank, world, unique_id):
        self._bind_library()
        self.h = _lib.handle(device)
        self.rank, self.world = int(rank), int(world)
        c = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), 128)
        self.h.check(self.h.lib.asr_comm_init(C.byref(c), self.h.h, self.rank, self.world, idbuf), 'asr_comm_init')
        self.c = c

    def allreduce_mean(self, flat):
        _chk(flat, torch.float32, 'buffer')
        if not flat.is_cuda or not flat.is_contiguous():
            raise ValueError('allreduce_mean: contiguous cuda tensor expected')
        self.h.check(self.h.lib.asr_allreduce_mean(self.c, _p(flat), flat.numel(), _s()), 'asr_allreduce_mean')
        return flat

    def close(self):
        if getattr(self, 'c', None):
            self.h.lib.asr_comm_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
