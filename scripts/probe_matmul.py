"""What the library GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this step's GEMM shapes, beside asr_gemm:
context for DESIGN's GEMM section (the product path uses the hand-written kernels of csrc/gemm.hip only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensorflow_end2end_speech_recognition_amd import ops

dev = torch.device('cuda:0')
shapes = [  # (name, M, N, K, form)
    ('xproj 5x256 B16', 12448, 2048, 512, 'nt'),
    ('xproj 5x512 B32 T1598', 51136, 4096, 1024, 'nt'),
    ('xproj cfgC', 105600, 4096, 1024, 'nt'),
    ('dW 5x256', 512, 1024, 12448, 'tn'),
    ('dW 5x512 cfgC', 1024, 2048, 105600, 'tn'),
]


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, M, N, K, form in shapes:
    fl = 2.0 * M * N * K
    if form == 'nt':
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev)
        outb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: torch.matmul(a, b.t(), out=outb)) if not os.environ.get('SKIP_LIB') else float('nan')
        au, bu = a, b
        t_own = timeit(lambda: ops.gemm(au, bu, transB=True, out=out))
        ref = torch.matmul(a[:4096].float(), b.float().t())
        err = (out[:4096] - ref).abs().max().item() / ref.abs().max().item()
        tail = (out[-300:] - torch.matmul(a[-300:].float(), b.float().t())).abs().max().item() / ref.abs().max().item()
        name = name + ' err %.1e/%.1e' % (err, tail)
    else:
        a = torch.randn(K, M, device=dev).bfloat16()
        b = torch.randn(K, N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev)
        outb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: torch.matmul(a.t(), b, out=outb))
        au, bu = a, b
        t_own = timeit(lambda: ops.gemm(au, bu, transA=True, out=out))
    print('%-44s M %6d N %5d K %6d  library (bf16 out) %8.1f us %6.0f TF/s | asr_gemm (fp32 out) %8.1f us %6.0f TF/s'
          % (name, M, N, K, t_lib * 1e6, fl / t_lib / 1e12, t_own * 1e6, fl / t_own / 1e12), flush=True)
