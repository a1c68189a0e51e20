import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
from tensorflow_end2end_speech_recognition_amd import ops
cuda = torch.device('cuda:0')
rng = np.random.RandomState(1)
for (N, H, W, Cin) in [(3, 40, 11, 3), (70, 5, 3, 3)]:
    Cout = 64
    x = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.float32).to(torch.bfloat16)
    w = torch.tensor(rng.randn(3, 3, Cin, Cout) * 0.2, dtype=torch.float32).to(torch.bfloat16)
    b = torch.tensor(rng.randn(Cout) * 0.1, dtype=torch.float32)
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), padding=1)
    ref = torch.relu(y64).permute(0, 2, 3, 1)
    got = ops.conv3x3_smallc_fwd(x.to(cuda), w.to(cuda).view(9 * Cin, Cout), b.to(cuda), relu=True).float().cpu().double()
    # the im2col + GEMM path for comparison
    patches = ops.im2col3x3(x.to(cuda), 32)
    alt = ops.gemm(patches[:, :9 * Cin], w.to(cuda).view(9 * Cin, Cout), bias=b.to(cuda), relu=True).float().cpu().double().view(N, H, W, Cout)
    for name, g in (('direct', got), ('im2col', alt)):
        err = (g - ref).abs()
        ulp = err / (ref.abs() * 2.0 ** -8 + 1e-6)
        print(N, H, W, name, 'max abs err %.3e  max err in bf16 ulps %.2f  frac > 0.51 ulp %.4f' % (err.max(), ulp.max(), (ulp > 0.51).double().mean()))
    print('direct vs im2col identical fraction', (got == alt).double().mean().item())
