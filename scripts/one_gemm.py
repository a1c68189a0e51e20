"""One GEMM shape in a loop (for rocprofv3 counter passes): xprojT [12448,512]x[1024,512]^T."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
A = torch.randn(12448, 512, device=dev).to(torch.bfloat16)
Bt = torch.randn(1024, 512, device=dev).to(torch.bfloat16)
out = torch.empty(12448, 1024, device=dev)
for _ in range(10):
    ops.gemm(A, Bt, transB=True, out=out)
torch.cuda.synchronize()
