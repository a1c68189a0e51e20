#!/usr/bin/env python
"""Which phase of the cluster recurrence stretches when other kernels run beside it?
One-layer 256-unit BLSTM step (headline batch) three ways: alone, beside a stream of weight-gradient-sized bf16 GEMMs,
beside a stream of memory-bound kernels (dropout-mask generation).  Prints the HIP-event duration of the forward / BPTT
launch and the in-kernel phase timers (ASR_LSTM_DBG=1: s_memtime ticks per step and phase, slowest wave of cluster 0).
usage: python scripts/probe_interference.py [out.json]"""
import ctypes
import json
import os
import sys
os.environ['ASR_LSTM_DBG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch   # noqa: E402
from bench import make_batch, KernelTimer   # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC   # noqa: E402
from tensorflow_end2end_speech_recognition_amd import _lib, ops   # noqa: E402

dev = torch.device('cuda:0')
x, sl, labels, dense = make_batch(1, 16, 120, 62, 100, 778)
m = CTC('blstm', 120, 256, 1, 61, dtype='bf16', seed=0)
xd = torch.tensor(x, device=dev)
sld = torch.tensor(sl, device=dev)
lib = _lib.load()
lib.asr_debug_cluster_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
A = torch.randn(12448, 512, device=dev).to(torch.bfloat16)
B = torch.randn(12448, 1024, device=dev).to(torch.bfloat16)
big = torch.empty(64 << 20, device=dev)
side = torch.cuda.Stream()
timer = KernelTimer(ops, ["lstm_fwd", "lstm_bwd"])
timer.install()


def phases():
    buf = (ctypes.c_ulonglong * 1280)()
    assert lib.asr_debug_cluster_cycles(buf, 1280) == 0
    allv = np.array(list(buf), dtype=np.float64)
    out = {}
    for label, a in (('fwd', allv[256:768].reshape(2, 4, 8, 8)), ('bwd', allv[768:1280].reshape(2, 4, 8, 8))):
        steps = a[0, 0, 0, 5]
        per = a[0, 0, :, :4] / max(steps, 1)                 # direction 0, CU 0: [wave, phase]
        w = int(per.sum(1).argmax())
        out[label] = dict(steps=steps, ticks_per_step=[round(v, 1) for v in per[w]], total=round(per[w].sum(), 1),
                          fast=a[0, 0, 0, 6], repolls_per_step=round(a[0, 0, w, 4] / max(steps, 1), 2))
    return out


def step(load):
    torch.cuda.synchronize()
    if load is not None:
        with torch.cuda.stream(side):
            for _ in range(60):
                load()
    timer.records = {n: [] for n in timer.names}
    timer.enabled = True
    loss, _ = m.compute_loss(xd, dense, sld, keep_prob=1.0, is_training=True)
    m.train(loss, 'sgd', 0.0)
    timer.enabled = False
    torch.cuda.synchronize()
    t = timer.summary() if hasattr(timer, 'summary') else {}
    return t, phases()


res = {}
for name, load in (('alone', None), ('beside bf16 GEMMs', lambda: ops.gemm(A, B, transA=True, out_dtype='f32')),
                   ('beside memory-bound kernels', lambda: ops.dropout_mask((64 << 20,), 0.8, 1, 0, dev)),
                   ('alone again', None)):
    step(load)
    t, ph = step(load)
    res[name] = dict(kernels=t, phases=ph)
    print(name, json.dumps(t), json.dumps(ph))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], 'w'), indent=1)
