"""The 3-channel first convolution: matrix-core kernels against the vector-ALU kernels and the fp64 convolution of the same
bf16 operands (run once per ASR_SMALLC_MFMA value; the second run compares with the first run's saved output)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensorflow_end2end_speech_recognition_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
N, H, W, Cin, Cout = 300, 40, 11, 3, 64
x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
w = (torch.randn(3, 3, Cin, Cout, generator=g) * 0.2).to(torch.bfloat16)
b = torch.randn(Cout, generator=g) * 0.1
dpre = torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16)
out = ops.conv3x3_smallc_fwd(x.to(dev), w.view(9 * Cin, Cout).to(dev), b.to(dev), relu=True)
dw = ops.conv3x3_smallc_bwd_weight(x.to(dev), dpre.to(dev), torch.zeros(9 * Cin, Cout, device=dev))
ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), padding=1)).permute(0, 2, 3, 1)
pat = F.unfold(x.double().permute(0, 3, 1, 2), 3, padding=1)          # [N, Cin*9, H*W] (ci-major, tap-minor)
pat = pat.view(N, Cin, 9, H * W).permute(0, 3, 2, 1).reshape(N * H * W, 9 * Cin)   # -> (tap, ci)
refw = pat.t() @ dpre.double().view(-1, Cout)
o = out.float().cpu().double()
refb = ref.to(torch.bfloat16).double()
tag = os.environ.get('ASR_SMALLC_MFMA', '1')
print('mfma=%s fwd: max|out - ref| %.3e  mean %.3e  elements != bf16(ref): %.4f %%' % (
    tag, (o - ref).abs().max(), (o - ref).abs().mean(), 100.0 * (o != refb).double().mean()))
print('mfma=%s wgrad: max rel err %.3e' % (tag, ((dw.cpu().double() - refw).abs().max() / refw.abs().max())))
path = '/tmp/smallc_%s.pt' % ('a' if tag == '0' else 'b')
torch.save(dict(out=out.cpu(), dw=dw.cpu()), path)
other = '/tmp/smallc_%s.pt' % ('b' if tag == '0' else 'a')
if os.path.exists(other):
    d = torch.load(other)
    print('vs the other form: fwd elements that differ %.4f %%, max |diff| %.3e; wgrad max rel diff %.3e' % (
        100.0 * (d['out'] != out.cpu()).double().mean(), (d['out'].float() - out.cpu().float()).abs().max(),
        (d['dw'] - dw.cpu()).abs().max() / dw.abs().max()))
