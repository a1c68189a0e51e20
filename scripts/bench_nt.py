"""Time the NT projection products (bf16 operands, fp32 out) of the cfg C / D shapes on the whole chip."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(104192, 1024, 4096), (104192, 4096, 1024), (51136, 1024, 4096), (51136, 4096, 1024), (12448, 512, 2048)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); Bt = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    us = t(lambda: ops.gemm(A, Bt, transB=True, out=out))
    ref = A[:64].float() @ Bt.float().t()
    err = (out[:64] - ref).abs().max().item() / ref.abs().max().item()
    ref2 = A[-64:].float() @ Bt.float().t()
    err2 = (out[-64:] - ref2).abs().max().item() / ref2.abs().max().item()
    fl = 2.0 * M * K * N
    print('NT M=%6d K=%4d N=%4d: %8.1f us %7.1f TF/s (rel err %.1e / %.1e)' % (M, K, N, us, fl / us / 1e6, err, err2), flush=True)
