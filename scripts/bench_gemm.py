"""Time the GEMM shapes of the headline step (TB = 12448 rows, 5x256 BLSTM) in isolation."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
TB = 12448
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def rnd(*s, dt=torch.bfloat16): return torch.randn(*s, device=dev).to(dt)
cases = []
for K in (120, 512):
    for N in (1024, 2048):
        A, Bm = rnd(TB, K), rnd(K, N); bias = torch.randn(N, device=dev)
        out = torch.empty(TB, N, device=dev)
        cases.append(('xproj   [%d,%d]x[%d,%d] ->f32' % (TB, K, K, N), lambda A=A, Bm=Bm, bias=bias, out=out: ops.gemm(A, Bm, bias=bias, out=out), 2 * TB * K * N, TB * N * 4 + TB * K * 2))
for K in (120, 512):
    A, Bt = rnd(TB, K), rnd(1024, K); bias = torch.randn(1024, device=dev)
    out = torch.empty(TB, 1024, device=dev)
    cases.append(('xprojT  [%d,%d]x[1024,%d]^T ->f32' % (TB, K, K), lambda A=A, Bt=Bt, bias=bias, out=out: ops.gemm(A, Bt, transB=True, bias=bias, out=out), 2 * TB * K * 1024, TB * 1024 * 4 + TB * K * 2))
for N in (120, 512):
    for K in (1024, 2048):
        dG, W = rnd(TB, K), rnd(N, K)
        out = torch.empty(TB, N, device=dev)
        cases.append(('dx      [%d,%d]x[%d,%d]^T ->f32' % (TB, K, N, K), lambda dG=dG, W=W, out=out: ops.gemm(dG, W, transB=True, out=out), 2 * TB * K * N, TB * N * 4 + TB * K * 2))
for M in (120, 512, 256):
    X, dG = rnd(TB, M), rnd(TB, 1024)
    out = torch.empty(M, 1024, device=dev)
    cases.append(('dW      [%d,%d]^T x[%d,1024] ->f32' % (TB, M, TB), lambda X=X, dG=dG, out=out: ops.gemm(X, dG, transA=True, out=out), 2 * TB * M * 1024, TB * (M + 1024) * 2))
A, W = rnd(TB, 512), rnd(512, 62); out = torch.empty(TB, 62, device=dev); bias = torch.randn(62, device=dev)
cases.append(('logits  [%d,512]x[512,62]' % TB, lambda: ops.gemm(A, W, bias=bias, out=out), 2 * TB * 512 * 62, TB * 512 * 2))
for name, fn, fl, by in cases:
    us = t(fn)
    print('%-44s %8.1f us  %7.1f TF/s  %7.1f GB/s(min traffic)' % (name, us, fl / us / 1e6, by / us / 1e3))
