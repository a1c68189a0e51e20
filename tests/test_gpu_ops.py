"""GPU parity tests: every HIP op through the C ABI vs the CPU oracle on the same seeded
inputs.  Tolerances: fp32 path 1e-4 relative (north_star), bf16 path documented looser."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decoders as odec
from oracle import lstm as olstm
from oracle import optim as oopt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _ops():
    from tensorflow_end2end_speech_recognition_amd import ops
    return ops


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(1, 1), (63, 65), (120, 1024), (512, 1024), (130, 70)])
def test_transpose2d(cuda, dtype, shape):
    x = torch.randn(shape[0], shape[1] + 3, device=cuda).to(dtype)[:, :shape[1]]   # strided rows
    y = _ops().transpose2d(x)
    assert y.shape == (shape[1], shape[0])
    assert torch.equal(y, x.t().contiguous())


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(5, 7, 5, 64, 64), (3, 40, 11, 64, 128), (70, 6, 3, 128, 128), (2, 1, 1, 64, 64),
                                            # N >= 64 images that fit the LDS: the image-resident kernels (conv3x3_img_kernel),
                                            # all four channel shapes (the data gradient runs the mirrored one), the cfg C
                                            # image sizes, more images than CUs (a workgroup walks several), 1 x 1 images
                                            (70, 40, 11, 64, 64), (66, 20, 6, 64, 128), (65, 20, 6, 128, 128),
                                            (300, 5, 4, 64, 64), (64, 1, 1, 128, 64)])
def test_conv3x3_implicit_gemm(cuda, N, H, W, Cin, Cout):
    """Implicit-GEMM 3x3 SAME convolution (forward, data gradient, weight gradient) against
    torch.nn.functional.conv2d in fp64 on the same bf16-rounded operands."""
    ops = _ops()
    rng = np.random.RandomState(N + H + Cin)
    x = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.bfloat16)
    w = torch.tensor(rng.randn(3, 3, Cin, Cout) * 0.05, dtype=torch.float32)
    b = torch.tensor(rng.randn(Cout), dtype=torch.float32)
    dy = torch.tensor(rng.randn(N, H, W, Cout), dtype=torch.bfloat16)
    wf, wb = ops.conv3x3_prep_weights(w.to(cuda))
    wq = wf.float().cpu().view(Cout, 3, 3, Cin).permute(1, 2, 3, 0).double()        # the bf16-rounded weights, HWIO
    assert torch.equal(wb.cpu(), wq.flip(0, 1).permute(2, 0, 1, 3).reshape(Cin, 9 * Cout).to(torch.bfloat16))
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)                           # NCHW
    wr = wq.permute(3, 2, 0, 1).clone().requires_grad_(True)                           # OIHW
    yr = torch.nn.functional.conv2d(xr, wr, b.double(), padding=1)
    out = ops.conv3x3_fwd(x.to(cuda), wf, b.to(cuda), relu=True)
    ref = torch.relu(yr).permute(0, 2, 3, 1).detach().numpy()
    assert np.abs(out.float().cpu().numpy() - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())
    (yr * dy.double().permute(0, 3, 1, 2)).sum().backward()
    dx = ops.conv3x3_bwd_data(dy.to(cuda), wb)
    assert _rel(dx.cpu().numpy(), xr.grad.permute(0, 2, 3, 1).numpy()) < 1e-5
    dw = torch.zeros(9 * Cin, Cout, device=cuda)
    ops.conv3x3_bwd_weight(x.to(cuda), dy.to(cuda), dw)
    refw = wr.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout).numpy()
    assert _rel(dw.cpu().numpy(), refw) < 1e-5
    ops.conv3x3_bwd_weight(x.to(cuda), dy.to(cuda), dw, accumulate=True)
    assert _rel(dw.cpu().numpy(), 2 * refw) < 1e-5
    # weight and bias gradient in one call (the bias sums from the dY images the weight-gradient kernel has staged):
    # bit-identical weights, bias = column sums of dY
    dw2 = torch.full((9 * Cin, Cout), 7.0, device=cuda)
    db = torch.full((Cout,), 7.0, device=cuda)
    ops.conv3x3_bwd_weight_bias(x.to(cuda), dy.to(cuda), dw2, db)
    dw1 = torch.zeros(9 * Cin, Cout, device=cuda)
    ops.conv3x3_bwd_weight(x.to(cuda), dy.to(cuda), dw1)
    assert torch.equal(dw1, dw2)
    refb = dy.double().sum(dim=(0, 1, 2)).numpy()
    assert np.abs(db.cpu().numpy() - refb).max() < 1e-5 * max(1.0, float(dy.double().abs().sum(dim=(0, 1, 2)).max()))


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (130, 70, 45), (1, 62, 512), (257, 1024, 120),
                                   (700, 130, 1000), (16, 16, 4), (3000, 2048, 120), (120, 1024, 5000),
                                   # M <= 32, K % 64 == 0, N % 32 == 0: the skinny kernel of the decoder steps
                                   (32, 2048, 1600), (32, 1600, 2048), (32, 128, 512), (17, 64, 128), (1, 96, 512)])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_f32(cuda, M, N, K, ta, tb):
    ops = _ops()
    rng = np.random.RandomState(M + N + K + ta * 2 + tb)
    A = rng.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    B = rng.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64) + bias
    out = ops.gemm(torch.tensor(A, device=cuda), torch.tensor(B, device=cuda), bool(ta), bool(tb),
                   bias=torch.tensor(bias, device=cuda))
    assert _rel(out.cpu().numpy(), ref) < 2e-6
    # accumulate + strided C view
    big = torch.ones((M, N + 8), dtype=torch.float32, device=cuda)
    ops.gemm(torch.tensor(A, device=cuda), torch.tensor(B, device=cuda), bool(ta), bool(tb),
             out=big[:, 4:4 + N], accumulate=True)
    assert _rel(big[:, 4:4 + N].cpu().numpy(), ref - bias + 1.0) < 2e-6
    assert float(big[:, :4].min()) == 1.0 and float(big[:, 4 + N:].max()) == 1.0


@pytest.mark.parametrize('M,N,K', [(64, 64, 64), (130, 70, 45), (257, 1024, 120), (3000, 2048, 512),
                                   (256, 1024, 12448), (120, 1024, 12448), (512, 2040, 5001), (1024, 64, 3000)])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_bf16(cuda, M, N, K, ta, tb):
    ops = _ops()
    rng = np.random.RandomState(M + N + K)
    A = torch.tensor(rng.randn(*((K, M) if ta else (M, K))), dtype=torch.bfloat16)
    B = torch.tensor(rng.randn(*((N, K) if tb else (K, N))), dtype=torch.bfloat16)
    Af, Bf = A.double().numpy(), B.double().numpy()
    ref = (Af.T if ta else Af) @ (Bf.T if tb else Bf)
    out = ops.gemm(A.to(cuda), B.to(cuda), bool(ta), bool(tb), out_dtype='f32')
    # inputs are exactly representable; only fp32 accumulation order differs
    assert _rel(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('M,N,K,dt', [(12448, 512, 2048, 'bf16'), (1500, 256, 128, 'bf16'), (130, 70, 45, 'bf16'),
                                        (257, 64, 120, 'f32')])
def test_gemm_with_multiplier_epilogue(cuda, M, N, K, dt):
    """asr_gemm_mul (the dropout mask of the layer below folded into the dx GEMM): bit-identical to the GEMM followed
    by asr_apply_mask, on the lean NT kernel (fused epilogue) and on the shapes that take the generic kernel + one
    multiply pass."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(M + N)
    tdt = torch.bfloat16 if dt == 'bf16' else torch.float32
    A = torch.randn(M, K, generator=g).to(cuda).to(tdt)
    Bt = torch.randn(N, K, generator=g).to(cuda).to(tdt)
    mask = (torch.rand(M, N, generator=g) < 0.8).float().to(cuda) / 0.8
    plain = ops.gemm(A, Bt, transB=True, out_dtype='f32')
    fused = ops.gemm(A, Bt, transB=True, out_dtype='f32', mul=mask)
    assert torch.equal(fused, ops.apply_mask(plain, mask))
    with pytest.raises(ValueError):
        ops.gemm(A, Bt, transB=True, out_dtype='f32', mul=mask[:, :N - 1])


@pytest.mark.parametrize('M,N,K', [(66000, 512, 128), (33003, 1024, 192)])
def test_gemm_nt_large_tiles(cuda, M, N, K):
    """Products of >= 512 tiles of 256 x 256 take the eight-wave 256 x 256 NT kernel (buffer-load staging, the last M tile
    ragged): plain, + bias, accumulate, fp32 multiplier, dropout formed in the epilogue, bf16 result -- each against the
    exact product of the bf16 operands (fp64) / bit-identical to the unfused form."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(cuda)
    Bt = torch.randn(N, K, generator=g).to(torch.bfloat16).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    ref = A.double() @ Bt.double().t()
    out = ops.gemm(A, Bt, transB=True, out_dtype='f32')
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) < 2e-6 * scale
    outb = ops.gemm(A, Bt, transB=True, out_dtype='f32', bias=bias)
    assert float((outb.double() - (ref + bias.double())).abs().max()) < 2e-6 * scale
    acc = out.clone()
    ops.gemm(A, Bt, transB=True, out=acc, accumulate=True)
    assert float((acc.double() - 2 * ref).abs().max()) < 4e-6 * scale
    mask = (torch.rand(M, N, generator=g) < 0.8).float().to(cuda) / 0.8
    assert torch.equal(ops.gemm(A, Bt, transB=True, out_dtype='f32', mul=mask), ops.apply_mask(out, mask))
    d = (0.8, 5, (2 << 32) + 7)
    assert torch.equal(ops.gemm(A, Bt, transB=True, out_dtype='f32', drop=d),
                       ops.apply_mask(out, ops.dropout_mask((M, N), *d, cuda)))
    o16 = ops.gemm(A, Bt, transB=True, out_dtype='bf16')
    assert float((o16.double() - ref).abs().max()) < 5e-3 * scale
    # the last rows (ragged tile) and nothing beyond the tensor
    guard = torch.full((M + 2, N), 7.0, device=cuda)
    ops.gemm(A, Bt, transB=True, out=guard[:M])
    assert torch.equal(guard[:M], out) and float(guard[M:].min()) == 7.0


def test_gemm_tn_xcd_skip_is_result_neutral(cuda):
    """The side lanes' reduction-major GEMM with the first n XCDs left alone (asr_set_xcd_skip): bit-identical to n = 0
    for every n, including shapes whose tile count is not a multiple of the XCDs in use."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(3)
    for (K, M, N) in [(12448, 512, 1024), (4096, 120, 1024), (3000, 256, 1032)]:
        A = torch.randn(K, M, generator=g).to(cuda).to(torch.bfloat16)
        B = torch.randn(K, N, generator=g).to(cuda).to(torch.bfloat16)
        outs = []
        for n in (0, 2, 4, 6):
            ops.set_side_xcd_skip(A.device, n)
            with ops.side_lane(A.device):
                outs.append(ops.gemm(A, B, transA=True, out_dtype='f32'))
            ops.join_side(A.device)
        ops.set_side_xcd_skip(A.device, 0)
        torch.cuda.synchronize()
        ref = A.float().t() @ B.float()
        assert (outs[0] - ref).abs().max() < 2e-3 * ref.abs().max()
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


def test_gemm_bad_args(cuda):
    ops = _ops()
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(4, 5, device=cuda), torch.zeros(6, 4, device=cuda))


# --------------------------------------------------------------------------- LSTM
def _lstm_case(rng, T, B, D, H, ndir, lens, init=0.3):
    x = rng.randn(B, T, D)
    for b in range(B):
        x[b, lens[b]:] = 0
    ps = [olstm.init_lstm_params(rng, D, H, init=init) for _ in range(ndir)]
    for p in ps:
        p['b'] = torch.tensor(rng.uniform(-init, init, 4 * H))
    return x, ps


def _run_hip_layer(cuda, x, ps, lens, H, ndir, dtype, cell_clip, dout=None, dfinal=None, saved_fill=None,
                   clip_no_grad=0.0):
    ops = _ops()
    from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16, ASR_F32
    dt = ASR_BF16 if dtype == 'bf16' else ASR_F32
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    B, T, D = x.shape
    xd = ops.bt_to_tb(torch.tensor(x, dtype=torch.float32, device=cuda), dt)
    xproj = torch.empty((T, B, ndir * 4 * H), dtype=torch.float32, device=cuda)
    whf = torch.empty((ndir, 4 * H * H), dtype=tdt, device=cuda)
    whb = torch.empty_like(whf)
    for d, p in enumerate(ps):
        kernel = torch.tensor(p['w'].detach().numpy(), dtype=torch.float32, device=cuda)
        bias = torch.tensor(p['b'].detach().numpy(), dtype=torch.float32, device=cuda)
        w = ops.lstm_prep_weights(kernel, bias, D, H, dt)
        ops.gemm(xd.view(T * B, D), w['wx_il'], bias=w['bias_il'],
                 out=xproj.view(T * B, -1)[:, d * 4 * H:(d + 1) * 4 * H])
        whf[d].copy_(w['pf'])
        whb[d].copy_(w['pb'])
    peep = torch.tensor(np.stack([np.stack([p['wci'].detach().numpy(), p['wcf'].detach().numpy(), p['wco'].detach().numpy()]) for p in ps]),
                        dtype=torch.float32, device=cuda)
    sl = torch.tensor(lens, dtype=torch.int32, device=cuda)
    gates, hout, cs, cf, hf = ops.lstm_fwd(xproj, whf, peep, sl, H, ndir, dt, 1.0, cell_clip)
    res = dict(hout=hout.float().cpu().numpy(), cs=cs.cpu().numpy(), cf=cf.cpu().numpy(), hf=hf.cpu().numpy())
    # saved activations, device layout [T,B,ndir,H,4] -> [ndir][T,B,4,H]
    res['gates'] = gates.float().view(T, B, ndir, H, 4).permute(2, 0, 1, 4, 3).contiguous().cpu().numpy()
    if saved_fill is not None:     # overwrite the saved activations of frames past each utterance's length
        pad = torch.arange(T, device=cuda).view(T, 1, 1) >= sl.view(1, B, 1)
        gates.masked_fill_(pad, saved_fill)
        cs.masked_fill_(pad, saved_fill)
    if dout is not None:
        dcf = dhf = None
        if dfinal is not None:
            dcf = torch.tensor(dfinal[0], dtype=torch.float32, device=cuda)
            dhf = torch.tensor(dfinal[1], dtype=torch.float32, device=cuda)
        dg, dpeep = ops.lstm_bwd(torch.tensor(dout, dtype=torch.float32, device=cuda), gates, cs, whb, peep, sl,
                                 H, ndir, dt, dcf, dhf, clip_no_grad=clip_no_grad)
        # device layout [T,B,ndir,H,4] (gates interleaved) -> gate-major [T,B,ndir*4H] for the checks
        res['dgates'] = dg.float().view(T, B, ndir, H, 4).permute(0, 1, 2, 4, 3).reshape(T, B, ndir * 4 * H).cpu().numpy()
        res['dpeep'] = dpeep.cpu().numpy()
    return res


def _oracle_layer(x, ps, lens, ndir, cell_clip, dout=None, dfinal=None, clip_blocks_gradient=False):
    xt = torch.tensor(x).transpose(0, 1).contiguous().requires_grad_(True)
    sl = torch.tensor(lens, dtype=torch.long)
    for p in ps:
        for k in ('w', 'b', 'wci', 'wcf', 'wco'):
            p[k] = p[k].detach().clone().requires_grad_(True)
    kw = dict(forget_bias=1.0, cell_clip=cell_clip, use_peephole=True, clip_blocks_gradient=clip_blocks_gradient)
    outs, fins = [], []
    for d, p in enumerate(ps):
        o, f = olstm.dynamic_rnn(xt, sl, p, reverse=(d == 1), **kw)
        outs.append(o)
        fins.append(f)
    out = torch.cat(outs, 2)
    res = dict(hout=out.detach().numpy(), cf=np.stack([f[0].detach().numpy() for f in fins]),
               hf=np.stack([f[1].detach().numpy() for f in fins]))
    if dout is not None:
        obj = (out * torch.tensor(dout)).sum()
        if dfinal is not None:
            for d in range(ndir):
                obj = obj + (fins[d][0] * torch.tensor(dfinal[0][d])).sum() + (fins[d][1] * torch.tensor(dfinal[1][d])).sum()
        obj.backward()
        res['dx'] = xt.grad.numpy()
        res['dw'] = [p['w'].grad.numpy() for p in ps]
        res['db'] = [p['b'].grad.numpy() for p in ps]
        res['dpeep'] = np.stack([np.stack([p['wci'].grad.numpy(), p['wcf'].grad.numpy(), p['wco'].grad.numpy()]) for p in ps])
    return res


LSTM_SHAPES = [(12, 16, 24, 64, 2), (37, 32, 120, 128, 2), (20, 16, 40, 256, 2), (9, 16, 12, 64, 1),
               (15, 16, 30, 192, 2)]


@pytest.mark.parametrize('T,B,D,H,ndir', LSTM_SHAPES)
def test_lstm_fwd_f32(cuda, T, B, D, H, ndir):
    rng = np.random.RandomState(T * 7 + H)
    lens = rng.randint(1, T + 1, size=B)
    lens[0] = T
    lens[-1] = 0 if B > 16 else lens[-1]
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens)
    got = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'f32', 0.0)
    ref = _oracle_layer(x, ps, lens, ndir, 0.0)
    assert np.abs(got['hout'] - ref['hout']).max() < 2e-5
    assert np.abs(got['cf'] - ref['cf']).max() < 5e-5
    assert np.abs(got['hf'] - ref['hf']).max() < 2e-5
    for b in range(B):   # padded frames exactly zero
        assert np.abs(got['hout'][lens[b]:, b]).max() == 0 if lens[b] < T else True


def test_lstm_fwd_cell_clip(cuda):
    rng = np.random.RandomState(5)
    T, B, D, H = 14, 16, 20, 64
    lens = rng.randint(3, T + 1, size=B)
    x, ps = _lstm_case(rng, T, B, D, H, 2, lens, init=1.5)
    got = _run_hip_layer(cuda, x * 3, ps, lens, H, 2, 'f32', 0.5)
    ref = _oracle_layer(x * 3, ps, lens, 2, 0.5)
    # saved cell states are only defined on valid frames (the padding is never written)
    cmax = max(np.abs(got['cs'][:lens[b], b]).max() for b in range(B))
    assert cmax <= 0.5 + 1e-6 and cmax > 0.49   # clip active
    assert np.abs(got['hout'] - ref['hout']).max() < 5e-5


@pytest.mark.parametrize('T,B,D,H,ndir', [(14, 16, 20, 64, 2), (14, 32, 20, 128, 2), (12, 32, 20, 256, 2), (10, 16, 20, 320, 1),
                                          (10, 16, 16, 512, 2), (11, 16, 20, 192, 2)])
def test_lstm_bwd_gradient_blocking_clip(cuda, T, B, D, H, ndir):
    """asr_lstm_bwd_ex, clip_no_grad: tf.contrib.rnn.LSTMCell (the num_proj cell, reference blstm.py:187-230) clamps the
    cell state with tf.clip_by_value -- a clamped state passes nothing to its gates or to c_prev, while the fused
    LSTMBlockCell gradient is straight-through.  Every fp32 BPTT kernel (single-CU at 64 / 192, the clusters at 128 / 256 /
    320 / 512) against the oracle's autograd through torch.clamp, an ACTIVE clip; the straight-through gradient of the same
    case is measurably different, so the comparison means something."""
    rng = np.random.RandomState(T * 13 + H)
    lens = rng.randint(2, T + 1, size=B)
    lens[0] = T
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.3 if H > 128 else 1.0)
    x = x * 3
    clip = 0.5
    dout = rng.randn(T, B, ndir * H)
    dfinal = (rng.randn(ndir, B, H) * 0.5, rng.randn(ndir, B, H) * 0.5)
    got = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'f32', clip, dout, dfinal, clip_no_grad=clip)
    thru = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'f32', clip, dout, dfinal)
    ref = _oracle_layer(x, ps, lens, ndir, clip, dout, dfinal, clip_blocks_gradient=True)
    clamped = np.mean([np.mean(np.abs(got['cs'][:lens[b], b]) >= clip) for b in range(B)])
    assert 0.02 < clamped < 0.9, clamped
    dg = got['dgates'].astype(np.float64)
    xt = np.transpose(x, (1, 0, 2))
    dx = np.zeros_like(xt)
    for d, p in enumerate(ps):
        g = dg[:, :, d * 4 * H:(d + 1) * 4 * H].reshape(T * B, 4 * H)
        dx += (g @ p['w'].detach().numpy()[:D].T).reshape(T, B, D)
        assert _rel(xt.reshape(T * B, D).T @ g, ref['dw'][d][:D]) < 1e-4
        assert _rel(got['dpeep'][d, 3:7].reshape(-1), ref['db'][d]) < 1e-4
    assert _rel(dx, ref['dx']) < 1e-4
    assert _rel(got['dpeep'][:, :3], ref['dpeep']) < 1e-4
    assert _rel(thru['dgates'], got['dgates']) > 1e-2      # the straight-through gradient is another one
    for b in range(B):
        if lens[b] < T:
            assert np.abs(got['dgates'][lens[b]:, b]).max() == 0
    # bf16 operands (the cluster kernels' CLIPZ instantiations at 256 / 320 / 512, the single-CU kernel elsewhere): the same
    # mask -- the blocked gradient, not the straight-through one.  A state within bf16 rounding of the clip may fall on the other
    # side in the bf16 forward, so the comparison is over the input gradient as a whole, at bf16's bar
    lo = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', clip, dout, dfinal, clip_no_grad=clip)
    dxl = thru_dx(lo, ps, xt, T, B, D, H)
    err_lo, err_thru = _rel(dxl, ref['dx']), _rel(thru_dx(thru, ps, xt, T, B, D, H), ref['dx'])
    print('bf16 blocked-clip gradient: %.3f of the largest entry from the oracle (straight-through: %.3f)' % (err_lo, err_thru))
    assert err_lo < 0.5 * err_thru, (err_lo, err_thru)   # (a strongly driven cell: the bf16 forward states drift, the mask does not)


def thru_dx(res, ps, xt, T, B, D, H):
    dg = res['dgates'].astype(np.float64)
    dx = np.zeros_like(xt)
    for d, p in enumerate(ps):
        dx += (dg[:, :, d * 4 * H:(d + 1) * 4 * H].reshape(T * B, 4 * H) @ p['w'].detach().numpy()[:D].T).reshape(T, B, D)
    return dx


@pytest.mark.parametrize('T,B,D,H,ndir', LSTM_SHAPES)
def test_lstm_bwd_f32(cuda, T, B, D, H, ndir):
    rng = np.random.RandomState(T * 11 + H)
    lens = rng.randint(1, T + 1, size=B)
    lens[0] = T
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens)
    dout = rng.randn(T, B, ndir * H)
    dfinal = (rng.randn(ndir, B, H) * 0.5, rng.randn(ndir, B, H) * 0.5)
    got = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'f32', 0.0, dout, dfinal)
    ref = _oracle_layer(x, ps, lens, ndir, 0.0, dout, dfinal)
    # rebuild the parameter/input gradients from dgates exactly as the host driver does
    dg = got['dgates'].astype(np.float64)                     # [T,B,ndir*4H]
    xt = np.transpose(x, (1, 0, 2))
    hout = got['hout'].astype(np.float64)
    dx = np.zeros_like(xt)
    for d, p in enumerate(ps):
        g = dg[:, :, d * 4 * H:(d + 1) * 4 * H].reshape(T * B, 4 * H)
        w = p['w'].detach().numpy()
        dwx = xt.reshape(T * B, D).T @ g
        hp = np.zeros((T, B, H))
        if d == 0:
            hp[1:] = hout[:-1, :, :H]
        else:
            hp[:-1] = hout[1:, :, H:2 * H]
        dwh = hp.reshape(T * B, H).T @ g
        dx += (g @ w[:D].T).reshape(T, B, D)
        assert _rel(np.concatenate([dwx, dwh], 0), ref['dw'][d]) < 1e-4
        assert _rel(g.sum(0), ref['db'][d]) < 1e-4
        assert _rel(got['dpeep'][d, 3:7].reshape(-1), ref['db'][d]) < 1e-4
    assert _rel(dx, ref['dx']) < 1e-4
    assert _rel(got['dpeep'][:, :3], ref['dpeep']) < 1e-4
    for b in range(B):
        if lens[b] < T:
            assert np.abs(got['dgates'][lens[b]:, b]).max() == 0


@pytest.mark.parametrize('T,B,D,H,ndir', [(20, 16, 40, 256, 2), (11, 16, 24, 512, 2), (13, 32, 24, 320, 1)])
def test_lstm_bf16_and_wide(cuda, T, B, D, H, ndir):
    """bf16 operand path (and the 8-wave H=512 / odd H=320 instantiations): outputs within
    bf16 rounding of the fp64 oracle (tolerance 3e-2 abs on O(1) activations)."""
    rng = np.random.RandomState(H + T)
    lens = rng.randint(1, T + 1, size=B)
    lens[0] = T
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    dout = rng.randn(T, B, ndir * H)
    ref = _oracle_layer(x, ps, lens, ndir, 0.0, dout)
    for dtype, tol in (('f32', 1e-4), ('bf16', 3e-2)):
        got = _run_hip_layer(cuda, x, ps, lens, H, ndir, dtype, 0.0, dout)
        assert np.abs(got['hout'] - ref['hout']).max() < tol
        assert _rel(got['dpeep'][:, :3], ref['dpeep']) < (1e-4 if dtype == 'f32' else 5e-2)


@pytest.mark.parametrize('H', [256, 512, 320])
def test_lstm_cluster_exchange_paths(cuda, H):
    """The multi-CU recurrence (bf16) must give bit-identical results whether the cluster's per-step exchange uses
    same-XCD plain stores or the placement-independent write-through form (flag bit 4), and whether the forward
    all-gather travels as 4-byte self-tagged words (default: the step tag rides in the always-zero top exponent bits of
    the two bf16 halves) or as 8-byte {step, payload} granules (flag bit 10), and whether the BPTT reduce-scatter uses
    the consumer-major paired slots (default) or one 16-byte slot per (source, tile) (flag bit 11 inverts the per-shape default); no hand-off may time
    out."""
    ops = _ops()
    rng = np.random.RandomState(7)
    T, B, D, ndir = 61, 32, 40, 2
    lens = rng.randint(1, T + 1, size=B)
    lens[0], lens[5] = T, 1
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    dout = rng.randn(T, B, ndir * H)
    ref = _oracle_layer(x, ps, lens, ndir, 50.0, dout)
    # both cluster shapes: H/32 CUs x four waves (default) and H/64 CUs x eight waves (flag bit 9 = 512); the two sum
    # the k-chunks of h W_h in different orders, so bit-identity is asserted per shape, between the exchange flavours
    for base in ((0, 512) if H != 320 else (0,)):
        res = []
        try:
            for flags in (base, base | 16, base | 1024, base | 1024 | 16, base | 2048, base | 2048 | 16):
                ops.debug_set_lstm_flags(flags)
                res.append(_run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', 50.0, dout))
                assert ops.check_async_errors(0) == 0
        finally:
            ops.debug_set_lstm_flags(0)
        for other in res[1:]:
            for k in ('hout', 'cf', 'hf', 'dgates', 'dpeep'):
                assert np.array_equal(res[0][k], other[k]), (base, k)
            for b in range(B):   # saved cell states are only defined on valid frames
                assert np.array_equal(res[0]['cs'][:lens[b], b], other['cs'][:lens[b], b])
        assert np.abs(res[0]['hout'] - ref['hout']).max() < 3e-2


@pytest.mark.parametrize('T,B,ndir,clip,H,base', [(301, 16, 2, 0.0, 128, 0), (150, 32, 1, 2.0, 128, 0),
                                                  (778, 16, 2, 50.0, 128, 0), (301, 16, 2, 0.0, 128, 512),
                                                  (150, 32, 2, 50.0, 256, 0), (97, 16, 2, 0.0, 320, 0),
                                                  (120, 32, 1, 2.0, 320, 0), (131, 16, 2, 50.0, 512, 0),
                                                  (64, 32, 1, 0.0, 512, 32),
                                                  # flag bit 12: the exact-fp32 MFMA kernels instead of the three-term split
                                                  (301, 16, 2, 0.0, 128, 4096), (778, 16, 2, 50.0, 128, 4096),
                                                  (150, 32, 2, 50.0, 256, 4096), (150, 32, 1, 2.0, 128, 4096 | 32)])
def test_lstm_cluster_f32_long_sequences(cuda, T, B, ndir, clip, H, base):
    """fp32 operands run on the cluster kernels at every registry width: H = 128 (BASELINE configs[0]) on four CUs x four
    waves (and, flag bit 9, two CUs x eight waves), H = 256 / 320 / 512 on H/32 CUs x four waves.  Forward values, final
    states and every gradient the BPTT kernel produces against the fp64 oracle over hundreds of hand-offs, ragged
    lengths, with and without the cell clip; both exchange flavours bit-identical; no hand-off may time out.
    Round 5: at H = 128 / 256 (four waves) the default kernels multiply on the bf16 matrix pipe with every fp32 operand
    split into three bf16 terms (lstm_*_cluster_f32s_kernel); flag bit 12 selects the exact-fp32 MFMA kernels -- both
    are held to the SAME bounds here."""
    ops = _ops()
    rng = np.random.RandomState(T + B)
    D = 24
    lens = rng.randint(1, T + 1, size=B)
    lens[0], lens[3] = T, 1
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    dout = rng.randn(T, B, ndir * H)
    dfinal = (rng.randn(ndir, B, H) * 0.5, rng.randn(ndir, B, H) * 0.5)
    res = []
    try:
        for flags in (base, base | 16):
            ops.debug_set_lstm_flags(flags)
            res.append(_run_hip_layer(cuda, x, ps, lens, H, ndir, 'f32', clip, dout, dfinal))
            assert ops.check_async_errors(0) == 0
    finally:
        ops.debug_set_lstm_flags(0)
    for k in ('hout', 'cf', 'hf', 'dgates', 'dpeep'):
        assert np.array_equal(res[0][k], res[1][k]), k
    got = res[0]
    ref = _oracle_layer(x, ps, lens, ndir, clip, dout, dfinal)
    assert np.abs(got['hout'] - ref['hout']).max() < 5e-5
    assert np.abs(got['cf'] - ref['cf']).max() < 2e-4
    assert np.abs(got['hf'] - ref['hf']).max() < 5e-5
    dg = got['dgates'].astype(np.float64)
    xt = np.transpose(x, (1, 0, 2))
    hout = got['hout'].astype(np.float64)
    dx = np.zeros_like(xt)
    for d, p in enumerate(ps):
        dd = d if ndir == 2 else 0
        g = dg[:, :, dd * 4 * H:(dd + 1) * 4 * H].reshape(T * B, 4 * H)
        w = p['w'].detach().numpy()
        hp = np.zeros((T, B, H))
        if d == 0:
            hp[1:] = hout[:-1, :, :H]
        else:
            hp[:-1] = hout[1:, :, H:2 * H]
        assert _rel(np.concatenate([xt.reshape(T * B, D).T @ g, hp.reshape(T * B, H).T @ g], 0), ref['dw'][d]) < 2e-4
        assert _rel(got['dpeep'][d, 3:7].reshape(-1), ref['db'][d]) < 2e-4
        dx += (g @ w[:D].T).reshape(T, B, D)
    assert _rel(dx, ref['dx']) < 2e-4
    assert _rel(got['dpeep'][:, :3], ref['dpeep']) < 2e-4
    for b in range(B):
        if lens[b] < T:
            assert np.abs(got['hout'][lens[b]:, b]).max() == 0
            assert np.abs(got['dgates'][lens[b]:, b]).max() == 0


@pytest.mark.parametrize('H,dtype,B', [(256, 'bf16', 16), (512, 'bf16', 32), (128, 'f32', 16), (64, 'f32', 16)])
def test_bptt_never_reads_saved_activations_of_padded_frames(cuda, H, dtype, B):
    """The forward kernels park the accesses of rows past their length, so the saved gates / cell states of padded frames
    are whatever the allocator handed out.  The only consumer is asr_lstm_bwd: with those positions
    set to NaN or to zero it must produce the same BITS (dgates, zero at padded frames, and the peephole / bias sums)."""
    ops = _ops()
    rng = np.random.RandomState(H + B)
    T, D, ndir = 21, 24, 2
    lens = rng.randint(1, T + 1, size=B)
    lens[0], lens[1] = T, 1
    if B > 16:
        lens[-1] = 0
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    dout = rng.randn(T, B, ndir * H)
    a = _run_hip_layer(cuda, x, ps, lens, H, ndir, dtype, 50.0, dout, saved_fill=float('nan'))
    b = _run_hip_layer(cuda, x, ps, lens, H, ndir, dtype, 50.0, dout, saved_fill=0.0)
    assert np.isfinite(a['dgates']).all() and np.isfinite(a['dpeep']).all()
    assert np.array_equal(a['dgates'], b['dgates']) and np.array_equal(a['dpeep'], b['dpeep'])
    valid = np.arange(T)[:, None] < lens[None, :]
    assert np.abs(a['dgates'][~valid]).max() == 0
    assert ops.check_async_errors(0) == 0


def test_nonfinite_hidden_state_is_reported_not_laundered(cuda):
    """The forward all-gather's 4-byte self-tagged words force two bits of every published pair (the step tag), so a NaN
    h would reach the peer CUs as a finite value while its owner keeps the NaN: the kernel ORs the published
    pairs and raises bit 2 of the sticky error word at the end of the launch; the blocking check reports it as a
    non-finite state, not as a hand-off timeout, and clears it."""
    from tensorflow_end2end_speech_recognition_amd import _lib
    ops = _ops()
    rng = np.random.RandomState(5)
    T, B, D, H, ndir = 10, 16, 24, 256, 2
    lens = rng.randint(4, T + 1, size=B)
    lens[3] = T
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    assert ops.check_async_errors(0) == 0
    _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', 0.0)
    assert ops.check_async_errors(0) == 0                    # finite inputs: nothing raised
    xn = x.copy()
    xn[2, 3, 5] = np.nan                                     # one NaN feature of a valid frame
    _run_hip_layer(cuda, xn, ps, lens, H, ndir, 'bf16', 0.0)
    with pytest.raises(_lib.AsrError, match='non-finite'):
        ops.check_async_errors(0)
    assert ops.check_async_errors(0) == 0                    # reported once, then cleared
    _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', 0.0)
    assert ops.check_async_errors(0) == 0


def test_cluster_handoff_timeout_is_reported(cuda):
    """A hand-off that times out must not go unnoticed: with the test-only flag (one member of every cluster leaves
    early, spin limit 2000 polls) the sticky error word is raised, the blocking check raises at the next sync point,
    and the non-blocking watch armed by every optimizer step raises within ErrorWatch.DEPTH steps of training."""
    from tensorflow_end2end_speech_recognition_amd import _lib
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    ops = _ops()
    rng = np.random.RandomState(3)
    T, B, D, H, ndir = 12, 16, 24, 256, 2
    lens = rng.randint(4, T + 1, size=B)
    x, ps = _lstm_case(rng, T, B, D, H, ndir, lens, init=0.1)
    assert ops.check_async_errors(0) == 0
    try:
        ops.debug_set_lstm_flags(64)
        _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', 0.0, rng.randn(T, B, ndir * H))
        with pytest.raises(_lib.AsrError):
            ops.check_async_errors(0)
        assert ops.check_async_errors(0) == 0          # reported once, then cleared
        model = CTC('blstm', D, H, 1, 9, dtype='bf16', seed=0)
        labels = np.full((B, 2), 1, dtype=np.int64)
        with pytest.raises(_lib.AsrError):
            for _ in range(ops.ErrorWatch.DEPTH + 1):
                loss, _ = model.compute_loss(x.astype(np.float32), labels, lens.astype(np.int32), keep_prob=1.0)
                model.train(loss, 'sgd', 0.0)
        # reported ONCE: two more faulty steps arm the non-blocking watch with copies of the word, then the blocking check
        # reports it -- the armed copies must not raise the same error again during the clean steps that follow (they did:
        # a spurious second AsrError up to DEPTH steps after the caller had restored its checkpoint)
        for _ in range(2):
            try:     # (either call may be the one that reports: the deferred label check looks at the error word first)
                loss, _ = model.compute_loss(x.astype(np.float32), labels, lens.astype(np.int32), keep_prob=1.0)
                model.train(loss, 'sgd', 0.0)
            except (_lib.AsrError, ValueError):
                pass
        ops.debug_set_lstm_flags(0)
        try:
            ops.check_async_errors(0)
        except _lib.AsrError:
            pass
        try:
            ops.flush_deferred_checks()                 # (the faulty steps' "labels fit the frames" counters saw NaN logits)
        except (ValueError, _lib.AsrError):
            pass
        model = CTC('blstm', D, H, 1, 9, dtype='bf16', seed=0)     # "restore the last checkpoint": the faulty steps' NaN
        for _ in range(ops.ErrorWatch.DEPTH + 2):                  # gradients went into the old weights (0 x NaN)
            loss, _ = model.compute_loss(x.astype(np.float32), labels, lens.astype(np.int32), keep_prob=1.0)
            model.train(loss, 'sgd', 0.0)
    finally:
        ops.debug_set_lstm_flags(0)
    torch.cuda.synchronize()
    try:
        ops.check_async_errors(0)                       # drain whatever the last launches left behind
    except _lib.AsrError:
        pass
    got = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', 0.0)      # and the kernels work again
    assert ops.check_async_errors(0) == 0
    ref = _oracle_layer(x, ps, lens, ndir, 0.0)
    assert np.abs(got['hout'] - ref['hout']).max() < 3e-2


def _err_stats(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    return d.max() / max(np.abs(ref).max(), 1e-30), d.mean() / max(np.abs(ref).mean(), 1e-30)


# (H, B, T, cell_clip): the shapes bench.py times (cfg B: H=256, B=16, T<=778), the cfg C/D width and H = 320
HEADLINE_LSTM = [(256, 16, 778, 50.0, 0), (256, 32, 300, 1.0, 0), (512, 16, 300, 50.0, 0), (512, 32, 778, 50.0, 0),
                 (320, 16, 250, 50.0, 0), (320, 32, 90, 1.0, 0),    # 320: the width of most of the reference's recipes (5 CUs)
                 # flag bit 9: clusters of H/64 CUs x eight waves (the round-2 form; default is H/32 CUs x four waves)
                 (256, 16, 778, 50.0, 512), (512, 32, 300, 1.0, 512)]


@pytest.fixture
def lstm_flags():
    from tensorflow_end2end_speech_recognition_amd import ops

    def set_flags(f):
        ops.debug_set_lstm_flags(f)
    yield set_flags
    ops.debug_set_lstm_flags(0)


@pytest.mark.parametrize('H,B,T,clip,flags', HEADLINE_LSTM)
def test_lstm_cluster_bf16_gradient_parity_headline_shapes(cuda, lstm_flags, H, B, T, clip, flags):
    """The multi-CU bf16 recurrence kernels (lstm_{fwd,bwd}_cluster8_kernel<256|512>, what bench.py times) against the
    oracle evaluated ON THE SAME bf16-rounded operands (oracle.lstm.layer_*_np with round_fn=bf16_round reproduces the
    kernels' rounding points: x, W, fed-back / emitted h, saved gates, dG entering dG.W_h^T), at the headline shapes:
    ragged lengths up to T = 778 (the double-buffered tag/parity exchange runs for the full sequence), ndir = 2,
    gradients entering through the outputs AND the final states.  Compared: hout, cs, c/h_final, saved gates, dgates,
    dW_x, dW_h, db, dx, dpeep.  What remains between device and oracle is fp32-vs-fp64 arithmetic plus the rare
    element whose fp32 value sits on a bf16 rounding boundary and rounds the other way (1 bf16 ulp = 2^-8 relative on
    that element): max-norm tolerances are therefore a few bf16 ulps of the largest entry, the mean error is held
    ~10-100x tighter than the bf16 step.  Bounds = 2-3x what was measured on MI355X (round 2, gpurun r02_c1): every
    max-abs error of hout / gates was EXACTLY one bf16 ulp (2^-8), mean abs 2e-5 (H=256) .. 9e-5 (H=512); dgates
    against the explicit BPTT fed with the device's own activations: max 3.4e-3 .. 6.1e-3 of the largest entry,
    mean 3e-4 .. 1e-3; dW_x 1.7e-3 .. 4.5e-3, dW_h 1.2e-3 .. 2.9e-3, dx 2.0e-3 .. 3.7e-3, dpeep / db 1e-3 .. 2.6e-3.
    Reference semantics: models/encoders/core/blstm.py:286-323."""
    ops = _ops()
    lstm_flags(flags)
    ndir, D = 2, 48
    rng = np.random.RandomState(H + B + T)
    lens = rng.randint(T // 3, T + 1, size=B)
    lens[0], lens[3] = T, 1
    if B > 16:
        lens[17], lens[20] = 0, T
    x = olstm.bf16_round(rng.randn(B, T, D))
    for b in range(B):
        x[b, lens[b]:] = 0
    ps = [olstm.init_lstm_params(rng, D, H, init=0.1) for _ in range(ndir)]
    for p in ps:
        p['b'] = torch.tensor(rng.uniform(-0.1, 0.1, 4 * H))
    dout = rng.randn(T, B, ndir * H)
    dfinal = (rng.randn(ndir, B, H) * 0.5, rng.randn(ndir, B, H) * 0.5)
    got = _run_hip_layer(cuda, x, ps, lens, H, ndir, 'bf16', clip, dout, dfinal)
    assert ops.check_async_errors(0) == 0
    xt = np.ascontiguousarray(np.transpose(x, (1, 0, 2)))
    R = olstm.bf16_round
    valid = (np.arange(T)[:, None] < lens[None, :])                   # [T,B]
    checks = []                                                       # (what, measured, bound)

    def chk(what, val, bound):
        checks.append((what, float(val), bound))

    for d in range(ndir):
        pn = {k: v.detach().numpy() for k, v in ps[d].items()}
        rev = d == 1
        tag = 'bw ' if rev else 'fw '
        f = olstm.layer_forward_np(xt, lens, pn, rev, 1.0, clip, True, round_fn=R)
        if clip < 10:
            assert np.abs(f['cs']).max() == clip                      # the clip is exercised
        sl = slice(d * H, (d + 1) * H)
        # ---- forward
        e = np.abs(got['hout'][:, :, sl] - f['hout'])
        chk(tag + 'hout max abs', e.max(), 2 ** -7)
        chk(tag + 'hout mean abs', e.mean(), 2e-4)
        assert np.abs(got['hout'][:, :, sl][~valid]).max() == 0     # padded frames exactly zero
        ecs = np.abs(got['cs'][:, :, sl] - f['cs'])[valid]
        chk(tag + 'cs max abs / max|cs|', ecs.max() / max(1.0, np.abs(f['cs']).max()), 2e-2)
        chk(tag + 'cs mean abs', ecs.mean(), 8e-4)
        eg = np.abs(got['gates'][d] - f['gates'])[valid]
        chk(tag + 'gates max abs', eg.max(), 2 ** -6)
        chk(tag + 'gates mean abs', eg.mean(), 3.5e-4)
        chk(tag + 'c_final max abs', np.abs(got['cf'][d] - f['c_final']).max(), 2e-2)
        chk(tag + 'h_final max abs', np.abs(got['hf'][d] - f['h_final']).max(), 5e-3)
        # ---- backward kernel in isolation: explicit BPTT fed with the DEVICE's saved activations
        dg_dev = got['dgates'][:, :, d * 4 * H:(d + 1) * 4 * H].reshape(T, B, 4, H)
        # (saved activations past an utterance's length are unspecified memory: mask them before the oracle multiplies)
        g_dev = np.where(valid[:, :, None, None], got['gates'][d], 0.0)
        c_dev = np.where(valid[:, :, None], got['cs'][:, :, sl], 0.0)
        bwd = olstm.layer_backward_np(dout[:, :, sl], g_dev, c_dev, lens, pn, rev, True,
                                      dfinal[0][d], dfinal[1][d], round_fn=R)
        mx, mean = _err_stats(dg_dev, bwd['dgates'])
        chk(tag + 'dgates|device activations max rel', mx, 1.5e-2)
        chk(tag + 'dgates|device activations mean rel', mean, 2.5e-3)
        assert np.abs(dg_dev[~valid]).max() == 0
        chk(tag + 'dpeep|device activations', _rel(got['dpeep'][d, :3], bwd['dpeep']), 4e-3)
        chk(tag + 'db|device activations', _rel(got['dpeep'][d, 3:7].reshape(-1), bwd['db']), 3e-3)
        # ---- whole chain against the oracle's own forward (device never consulted)
        full = olstm.layer_backward_np(dout[:, :, sl], f['gates'], f['cs'], lens, pn, rev, True,
                                       dfinal[0][d], dfinal[1][d], round_fn=R)
        mx, mean = _err_stats(dg_dev, full['dgates'])
        chk(tag + 'dgates max rel', mx, 2e-2)
        chk(tag + 'dgates mean rel', mean, 6e-3)
        chk(tag + 'dpeep', _rel(got['dpeep'][d, :3], full['dpeep']), 6e-3)
        chk(tag + 'db', _rel(got['dpeep'][d, 3:7].reshape(-1), full['db']), 6e-3)
        # weight / input gradients as the host driver forms them from dgates (x^T dG, h_prev^T dG, dG W_x^T)
        dw_ref, dx_ref = olstm.layer_param_grads_np(xt, f['hout'], full['dgates'], lens, pn, rev, round_fn=R)
        dw_dev, dx_dev = olstm.layer_param_grads_np(xt, got['hout'][:, :, sl].astype(np.float64), dg_dev.astype(np.float64),
                                                    lens, pn, rev, round_fn=R)
        chk(tag + 'dW_x', _rel(dw_dev[:D], dw_ref[:D]), 1e-2)
        chk(tag + 'dW_h', _rel(dw_dev[D:], dw_ref[D:]), 7e-3)
        chk(tag + 'dx', _rel(dx_dev, dx_ref), 1e-2)
    table = '\n'.join('%-44s %.3e  (bound %.1e)%s' % (w, v, bnd, '' if v <= bnd else '   <-- FAIL') for w, v, bnd in checks)
    print('\nH=%d B=%d T=%d clip=%g\n%s' % (H, B, T, clip, table))
    assert all(v <= bnd for _, v, bnd in checks), table


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('din,H,ndir,peep,ldk', [(120, 256, 2, True, 128), (512, 256, 2, True, 512),
                                                 (40, 64, 1, False, 40), (24, 128, 2, False, 64)])
def test_lstm_prep_layer_and_grad_finish_match_the_per_direction_ops(cuda, dtype, din, H, ndir, peep, ldk):
    """asr_lstm_prep_layer (all weight images of a layer, one launch) against asr_lstm_prep_weights + asr_transpose2d +
    concatenation, bit for bit; asr_lstm_grad_finish against asr_gate_deinterleave + copies."""
    ops = _ops()
    from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16, ASR_F32
    dt = ASR_BF16 if dtype == 'bf16' else ASR_F32
    g = torch.Generator(device='cpu').manual_seed(din + H)
    mk = lambda *s: torch.randn(*s, generator=g).to(cuda)
    variables = []
    for d in range(ndir):
        row = [mk(din + H, 4 * H), mk(4 * H)] + ([mk(H), mk(H), mk(H)] if peep else [None, None, None])
        variables.append(tuple(row))
    w = ops.lstm_prep_layer(variables, din, H, dt, ldk=ldk)
    for d, v in enumerate(variables):
        old = ops.lstm_prep_weights(v[0], v[1], din, H, dt)
        G = 4 * H
        assert torch.equal(w['wx_cat'][:, d * G:(d + 1) * G], old['wx_il'])
        assert torch.equal(w['wxT'][d * G:(d + 1) * G, :din], ops.transpose2d(old['wx_il']))
        assert not w['wxT'][d * G:(d + 1) * G, din:].float().abs().sum().item()
        assert torch.equal(w['bias'][d * G:(d + 1) * G], old['bias_il'])
        assert torch.equal(w['whf'][d], old['pf']) and torch.equal(w['whb'][d], old['pb'])
        if peep:
            assert torch.equal(w['peep'][d], torch.stack(v[2:5]))
    assert (w['peep'] is None) == (not peep)
    # the way back
    dw_il = mk(ndir, din + H, 4 * H)
    dpeep = mk(ndir, 7, H)
    grads = []
    for d in range(ndir):
        row = [torch.zeros(din + H, 4 * H, device=cuda), torch.zeros(4 * H, device=cuda)]
        row += [torch.zeros(H, device=cuda) for _ in range(3)] if peep else [None, None, None]
        grads.append(tuple(row))
    ops.lstm_grad_finish(grads, dw_il, dpeep, H)
    for d in range(ndir):
        ref = ops.gate_deinterleave(dw_il[d], torch.zeros(din + H, 4 * H, device=cuda), H)
        assert torch.equal(grads[d][0], ref)
        assert torch.equal(grads[d][1], dpeep[d, 3:7].reshape(-1))
        if peep:
            for k in range(3):
                assert torch.equal(grads[d][2 + k], dpeep[d, k])


def test_bt_to_tb_padded_rows(cuda):
    ops = _ops()
    from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16, ASR_F32
    x = torch.randn(5, 7, 120, device=cuda)
    for dt, tdt in ((ASR_F32, torch.float32), (ASR_BF16, torch.bfloat16)):
        y = ops.bt_to_tb(x, dt, ld=128)
        assert y.shape == (7, 5, 128) and y.dtype == tdt
        assert torch.equal(y[:, :, :120], x.transpose(0, 1).to(tdt)) and not y[:, :, 120:].float().abs().sum().item()
        assert torch.equal(ops.bt_to_tb(x, dt), x.transpose(0, 1).to(tdt).contiguous())


@pytest.mark.parametrize('N,H,W,Cin,kh,kw,sh,sw', [(5, 40, 11, 3, 11, 21, 3, 2), (4, 14, 6, 32, 11, 11, 1, 2),
                                                   (3, 14, 3, 32, 3, 3, 1, 1), (2, 7, 5, 3, 11, 21, 3, 2), (3, 9, 8, 4, 2, 4, 2, 3)])
def test_im2col_col2im_any_kernel_and_stride(cuda, N, H, W, Cin, kh, kw, sh, sw):
    """asr_im2col / asr_col2im (SAME convolution of any kernel / stride as a GEMM: the CLDNN front-end) against
    torch's conv2d with TensorFlow's SAME padding made explicit: forward through a GEMM with random filters, input
    gradient through col2im; bf16 patches are the bf16 rounding of the fp32 ones."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(N * H + kw)
    x = torch.randn(N, H, W, Cin, generator=g)
    wgt = torch.randn(kh, kw, Cin, 6, generator=g)
    Ho, Wo = ops.conv_out_hw(H, W, sh, sw)
    ph, pw = max((Ho - 1) * sh + kh - H, 0), max((Wo - 1) * sw + kw - W, 0)
    xt = x.permute(0, 3, 1, 2).double().clone().requires_grad_(True)
    xp = torch.nn.functional.pad(xt, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    ref = torch.nn.functional.conv2d(xp, wgt.permute(3, 2, 0, 1).double(), stride=(sh, sw))        # [N,6,Ho,Wo]
    assert ref.shape[2:] == (Ho, Wo)
    K = kh * kw * Cin
    patches = ops.im2col(x.to(cuda), kh, kw, sh, sw, ldp=(K + 7) // 8 * 8)
    assert patches.shape == (N * Ho * Wo, (K + 7) // 8 * 8) and not patches[:, K:].abs().sum().item()
    out = patches[:, :K].double().cpu() @ wgt.reshape(K, 6).double()
    assert (out.view(N, Ho, Wo, 6).permute(0, 3, 1, 2) - ref.detach()).abs().max() < 1e-5
    pb = ops.im2col(x.to(cuda).to(torch.bfloat16), kh, kw, sh, sw)
    assert torch.equal(pb, patches[:, :K].to(torch.bfloat16))
    dout = torch.randn(N * Ho * Wo, 6, generator=g)
    ref.backward(dout.view(N, Ho, Wo, 6).permute(0, 3, 1, 2).double())
    dpat = (dout.double() @ wgt.reshape(K, 6).double().t()).float()
    din = ops.col2im(dpat.to(cuda), N, H, W, Cin, kh, kw, sh, sw)
    assert (din.cpu().double() - xt.grad.permute(0, 2, 3, 1)).abs().max() < 1e-4


# --------------------------------------------------------------------------- CTC
def _ctc_case(rng, T, B, C, lmax, scale=2.0):
    logits = (rng.randn(T, B, C) * scale).astype(np.float32)
    sl = rng.randint(max(1, T // 3), T + 1, size=B).astype(np.int32)
    sl[0] = T
    labs = []
    for b in range(B):
        L = rng.randint(0, min(lmax, sl[b] // 2) + 1)
        labs.append([int(v) for v in rng.randint(0, C - 1, size=L)])
    return logits, sl, labs


def _flat(labs):
    flat = np.asarray(sum(labs, []) or [0], dtype=np.int32)
    off = np.zeros(len(labs) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(l) for l in labs])
    return flat, off


# the last three: cfg E (CSJ kanji, C = 3386, labels up to ~166 at T = 1000) and the word-level vocabularies of
# examples/librispeech/training/train_ctc.py (18 641 / 26 642 classes + blank) -- the gradient kernel keeps a C-bit
# membership map, never a per-class accumulator, in LDS
@pytest.mark.parametrize('T,B,C,lmax', [(30, 4, 6, 8), (120, 16, 40, 30), (300, 16, 62, 75), (50, 3, 29, 20),
                                         (64, 2, 700, 25), (400, 2, 29, 150), (200, 3, 3386, 60), (500, 2, 3386, 166),
                                         (40, 3, 26643, 12),
                                         # extended label sequences of 2L+1 = 401 / 601 / 901 / 1101 / 1901 states:
                                         # 8, 12, 16, 24 and 32 states per lane of the single-wave recursion (3 and 6
                                         # are covered above)
                                         (700, 2, 29, 200), (800, 2, 29, 300), (1100, 2, 29, 450), (1300, 2, 29, 550),
                                         (2000, 1, 12, 950)])
def test_ctc_loss_and_grad(cuda, T, B, C, lmax):
    ops = _ops()
    rng = np.random.RandomState(T + C)
    logits, sl, labs = _ctc_case(rng, T, B, C, lmax)
    labs[0] = labs[0][:2] + labs[0][:2] if len(labs[0]) >= 2 else labs[0]   # force repeats
    if lmax >= 200:   # the long-label cases: one utterance really has lmax labels (sl[0] == T leaves room for repeats)
        labs[0] = [int(v) for v in rng.randint(0, C - 1, size=lmax)]
    if len(labs[-1]) >= 3:
        labs[-1][2] = labs[-1][0]                                            # a non-adjacent repeat of a class
    flat, off = _flat(labs)
    ref_loss, ref_grad = octc.ctc_loss_batch(logits.astype(np.float64), labs, sl)
    loss, grad, ninf = ops.ctc_loss(torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda),
                                    torch.tensor(off, device=cuda), torch.tensor(sl, device=cuda),
                                    max(len(l) for l in labs), grad_scale=1.0)
    assert np.abs(loss.cpu().numpy() - ref_loss).max() / max(ref_loss.max(), 1) < 1e-5
    assert np.abs(grad.cpu().numpy() - ref_grad).max() < 2e-5
    assert int(ninf.item()) == 0


@pytest.mark.parametrize('T,C,lmax', [(300, 62, 100), (700, 29, 200), (800, 29, 300), (1100, 29, 450), (1300, 29, 550),
                                      (2000, 12, 950)])
def test_ctc_multi_wave_recursion_is_bit_identical_to_one_wave(cuda, T, C, lmax):
    """Round 6: with 6 or more states per lane the alpha / beta recursions run on 2 or 4 waves (K / NW states per lane, the
    two states that cross a wave boundary through LDS behind one barrier per frame).  Same operands and operations per
    state: losses and gradients must be the SAME BITS as the one-wave recursion (ASR_CTC_WAVES=1), for every class of
    states per lane (6, 8, 12, 16, 24, 32), with ragged utterances (the short ones leave whole waves without a state)."""
    import os
    ops = _ops()
    rng = np.random.RandomState(T + lmax)
    B = 3
    logits = (rng.randn(T, B, C) * 2.0).astype(np.float32)
    sl = np.array([T, max(8, T // 3), max(8, (2 * T) // 3)], dtype=np.int32)
    labs = [[int(v) for v in rng.randint(0, C - 1, size=n)] for n in (lmax, 3, max(1, lmax // 2))]
    flat, off = _flat(labs)
    args = (torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda), torch.tensor(off, device=cuda),
            torch.tensor(sl, device=cuda), lmax)
    old = os.environ.get('ASR_CTC_WAVES')
    try:
        os.environ['ASR_CTC_WAVES'] = '1'
        l1, g1, n1 = ops.ctc_loss(*args, grad_scale=1.0)
        l1, g1 = l1.cpu().numpy().copy(), g1.cpu().numpy().copy()
        os.environ.pop('ASR_CTC_WAVES')
        l4, g4, n4 = ops.ctc_loss(*args, grad_scale=1.0)
    finally:
        if old is None:
            os.environ.pop('ASR_CTC_WAVES', None)
        else:
            os.environ['ASR_CTC_WAVES'] = old
    assert int(n1.item()) == 0 and int(n4.item()) == 0
    assert np.array_equal(l1.view(np.uint32), l4.cpu().numpy().view(np.uint32))
    assert np.array_equal(g1.view(np.uint32), g4.cpu().numpy().view(np.uint32))
    ref_loss, _ = octc.ctc_loss_batch(logits.astype(np.float64), labs, sl)
    assert np.abs(l1 - ref_loss).max() / ref_loss.max() < 1e-5


def test_ctc_peaked_posteriors_and_long_utterances(cuda):
    """The scaled linear-domain recursion over inputs where log-domain and linear-domain arithmetic differ most:
    near one-hot posteriors (logit gaps of 60: most emission probabilities sit at the 2^-126 floor), a long utterance
    whose likelihood is ~e^-2500 (far below fp64's smallest power of two without the per-8-frames renormalisation)
    and an utterance whose correct path runs through improbable frames."""
    ops = _ops()
    rng = np.random.RandomState(5)
    T, B, C = 1500, 3, 30
    logits = (rng.randn(T, B, C) * 1.5).astype(np.float32)
    labs = [[int(v) for v in rng.randint(0, C - 1, size=n)] for n in (180, 40, 75)]
    sl = np.array([T, 600, 900], dtype=np.int32)
    # utterance 1: peaked posteriors along a valid alignment (7 frames per label, blanks in between)
    for t in range(600):
        k = labs[1][min(t // 15, len(labs[1]) - 1)] if (t // 7) % 2 == 0 else C - 1
        logits[t, 1, :] = -30.0
        logits[t, 1, k] = 30.0
    logits[300:330, 1, :] = rng.randn(30, C).astype(np.float32)      # a stretch the model is unsure about
    # utterance 2: every label is improbable at every frame (gap 25 to the blank)
    logits[:, 2, C - 1] += 25.0
    flat, off = _flat(labs)
    ref_loss, ref_grad = octc.ctc_loss_batch(logits.astype(np.float64), labs, sl)
    assert ref_loss[0] > 1000 and ref_loss[2] > 1000
    loss, grad, ninf = ops.ctc_loss(torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda),
                                    torch.tensor(off, device=cuda), torch.tensor(sl, device=cuda), 180, 1.0)
    assert int(ninf.item()) == 0
    assert np.abs(loss.cpu().numpy() - ref_loss).max() / ref_loss.max() < 1e-5
    assert np.abs(grad.cpu().numpy() - ref_grad).max() < 2e-5


def test_ctc_matches_tensorflow_known_answer_vectors(cuda):
    """asr_ctc_loss against the constants of TensorFlow's own ctc_loss_op_test.py testBasic
    (tests/golden/tf_known_answers.py): two utterances, depth 6, 5 valid of 7 padded frames -- both losses and all
    60 gradient entries, i.e. the HIP kernel agrees with tf.nn.ctc_loss itself, not only with the oracle."""
    import sys
    sys.path.insert(0, GOLD)
    import tf_known_answers as tfk
    ops = _ops()
    logits = np.zeros((7, 2, tfk.CTC_DEPTH), dtype=np.float32)
    for b, (probs, _, _, _) in enumerate(tfk.CTC_CASES):
        logits[:5, b] = np.log(probs)
    labs = [list(c[1]) for c in tfk.CTC_CASES]
    flat, off = _flat(labs)
    sl = np.array([5, 5], dtype=np.int32)
    loss, grad, ninf = ops.ctc_loss(torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda),
                                    torch.tensor(off, device=cuda), torch.tensor(sl, device=cuda), 5, 1.0)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.abs(loss - [tfk.CTC_LOSS_0, tfk.CTC_LOSS_1]).max() < 1e-4
    assert np.abs(grad[:5, 0] - tfk.CTC_GRAD_0).max() < 2e-5 and np.abs(grad[:5, 1] - tfk.CTC_GRAD_1).max() < 2e-5
    assert not grad[5:].any() and int(ninf.item()) == 0


def test_ctc_edge_cases(cuda):
    """empty label rows, T == exact minimum, infeasible rows (-> 0 loss/grad), seq_len 0."""
    ops = _ops()
    rng = np.random.RandomState(0)
    T, B, C = 12, 5, 7
    logits = rng.randn(T, B, C).astype(np.float32)
    sl = np.array([12, 5, 3, 0, 7], dtype=np.int32)
    labs = [[], [1, 1, 1], [2, 2], [3], [0, 1, 2, 3, 4, 5, 0]]   # row1 needs exactly 5; row2 needs 3 has 3
    labs[2] = [2, 2, 2]                                          # needs 5 > 3 -> infeasible
    flat, off = _flat(labs)
    ref_loss, ref_grad = octc.ctc_loss_batch(logits.astype(np.float64), labs, sl)
    loss, grad, ninf = ops.ctc_loss(torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda),
                                    torch.tensor(off, device=cuda), torch.tensor(sl, device=cuda), 7, 1.0)
    assert np.abs(loss.cpu().numpy() - ref_loss).max() < 1e-4
    assert np.abs(grad.cpu().numpy() - ref_grad).max() < 2e-5
    assert loss[2].item() == 0 and float(grad[:, 2].abs().max()) == 0 and int(ninf.item()) == 1
    assert loss[3].item() == 0 and float(grad[:, 3].abs().max()) == 0
    # deterministic
    loss2, grad2, _ = ops.ctc_loss(torch.tensor(logits, device=cuda), torch.tensor(flat, device=cuda),
                                   torch.tensor(off, device=cuda), torch.tensor(sl, device=cuda), 7, 1.0)
    assert torch.equal(grad, grad2) and torch.equal(loss, loss2)


def test_greedy_decode_bit_exact(cuda):
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'decoders_v1.npz'))
    for i in range(int(g['num_cases'])):
        probs, sl = g['c%d_probs' % i], g['c%d_seq_len' % i]
        C = probs.shape[2]
        logits = np.log(probs).astype(np.float32).transpose(1, 0, 2).copy()     # [T,1,C]
        lab, n = ops.ctc_greedy_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda))
        ref = odec.greedy_decode(logits.transpose(1, 0, 2), sl, C - 1)[0]        # same fp32 values
        assert lab[0, :int(n[0])].cpu().tolist() == ref
        assert lab[0, int(n[0]):].cpu().tolist() == [-1] * (logits.shape[0] - int(n[0]))
    rng = np.random.RandomState(1)
    T, B, C = 700, 16, 62
    logits = rng.randn(T, B, C).astype(np.float32)
    logits[:, :, C - 1] += 2.0
    logits[5:9] = logits[4]           # runs of identical frames (ties on repeats)
    logits[20, :, 3] = logits[20, :, 7] = 50.0    # exact tie -> lowest index wins
    sl = rng.randint(0, T + 1, size=B).astype(np.int32)
    sl[0], sl[1] = T, 0
    lab, n = ops.ctc_greedy_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda))
    ref = odec.greedy_decode(logits.transpose(1, 0, 2), sl, C - 1)
    for b in range(B):
        assert lab[b, :int(n[b])].cpu().tolist() == ref[b]


def test_beam_search_matches_reference_golden(cuda):
    """Every golden case of the reference's numpy BeamSearchDecoder (beam widths 1..20): labels
    bit-exact, -log score within 1e-5 (the kernel consumes fp32 logits = log of the golden fp64
    posteriors; the search itself is fp64)."""
    ops = _ops()
    g = np.load(os.path.join(GOLD, 'decoders_v1.npz'))
    checked = 0
    for i in range(int(g['num_cases'])):
        probs, sl = g['c%d_probs' % i], g['c%d_seq_len' % i]
        C = probs.shape[2]
        logits = np.log(probs).astype(np.float32).transpose(1, 0, 2).copy()     # [T,1,C]
        for w in g['c%d_widths' % i]:
            lab, n, score = ops.ctc_beam_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda), int(w))
            ref = list(g['c%d_beam%d' % (i, w)])
            assert lab[0, :int(n[0])].cpu().tolist() == ref, (i, int(w))
            assert abs(score[0].item() - float(g['c%d_beam%d_score' % (i, w)])) < 1e-5 * max(1, abs(score[0].item()))
            checked += 1
    assert checked >= 40


def test_beam_search_matches_tensorflow_known_answer(cuda):
    """asr_ctc_beam_decode and the class surface CTC.decoder(beam_width=2) -- the stand-in for the reference's
    tf.nn.ctc_beam_search_decoder call (models/ctc/ctc.py:344-346) -- on TensorFlow's own ctc_decoder_ops_test.py
    testCTCDecoderBeamSearch case (tests/golden/tf_known_answers.py): best beam [1, 0], its TF1 log_probability
    0.584855 once the frame-max normaliser is added to the device's -log p; logits carry the test's +2.0 offset and
    its frames beyond seq_len."""
    import sys
    sys.path.insert(0, GOLD)
    import tf_known_answers as tfk
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import sparsetensor2list
    ops = _ops()
    C = tfk.BEAM_PROBS.shape[1]
    logits = np.zeros((tfk.BEAM_PADDED_FRAMES, 1, C), dtype=np.float32)
    logits[:tfk.BEAM_PROBS.shape[0], 0] = np.log(tfk.BEAM_PROBS) + tfk.BEAM_LOGIT_OFFSET
    lg = torch.tensor(logits, device=cuda)
    sl = torch.tensor([tfk.BEAM_SEQ_LEN], dtype=torch.int32, device=cuda)
    lab, n, score = ops.ctc_beam_decode(lg, sl, tfk.BEAM_WIDTH)
    assert lab[0, :int(n[0])].cpu().tolist() == tfk.BEAM_DECODED[0]
    norm = tfk.beam_max_normaliser(tfk.BEAM_PROBS, tfk.BEAM_SEQ_LEN)
    assert abs(-score[0].item() + norm - tfk.BEAM_LOG_PROB[0]) < 5e-6
    model = CTC('blstm', 12, 64, 1, C - 1, dtype='f32', seed=0)
    for merge in (True, False):
        hyp = sparsetensor2list(model.decoder(lg, [tfk.BEAM_SEQ_LEN], beam_width=tfk.BEAM_WIDTH, merge_repeated=merge), 1)
        assert [int(v) for v in hyp[0]] == tfk.BEAM_DECODED[0]
    # a best path WITH repeats: the worked example of TensorFlow's documentation of the op, `A B B * B * B` ->
    # `A B` under merge_repeated=True (the reference's call, ctc.py:344-346), `A B B B` under False
    doc = torch.tensor(np.log(tfk.merge_doc_probs())[:, None].astype(np.float32), device=cuda)
    model3 = CTC('blstm', 12, 64, 1, tfk.MERGE_DOC_DEPTH - 1, dtype='f32', seed=0)
    for width in (2, 4, 8):
        for merge, want in ((True, tfk.MERGE_DOC_MERGED), (False, tfk.MERGE_DOC_UNMERGED)):
            hyp = sparsetensor2list(model3.decoder(doc, [doc.shape[0]], beam_width=width, merge_repeated=merge), 1)
            assert [int(v) for v in hyp[0]] == want, (width, merge, hyp)


def test_beam_search_cfgE_scale_matches_reference_golden(cuda):
    """BASELINE cfg E scale (C = 3387 classes, beam 20 and 100): labels bit-exact against the outputs of the
    REFERENCE'S OWN BeamSearchDecoder (tests/golden/decoders_cfge_v1.json, generated by make_golden_cfge.py in the
    build container; the seeded inputs are regenerated here and checked by digest).  Includes posteriors on a coarse
    grid (many exact ties at the class-pruning threshold) and nearly flat posteriors (no class is prunable)."""
    import json
    import sys
    sys.path.insert(0, GOLD)
    import cfge_inputs
    ops = _ops()
    gold = json.load(open(os.path.join(GOLD, 'decoders_cfge_v1.json')))
    assert len(gold) >= 4
    for name, g in sorted(gold.items()):
        probs, sl, W = cfge_inputs.posteriors(name)
        logits = cfge_inputs.fp32_logits(probs)
        assert cfge_inputs.digest(logits) == g['logits_sha256'], name
        assert W == g['beam_width'] and probs.shape[2] == g['C']
        lab, n, score = ops.ctc_beam_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda), W)
        assert lab[0, :int(n[0])].cpu().tolist() == g['labels'], name
        assert abs(score[0].item() - g['score']) < 1e-5 * max(1.0, abs(g['score'])), name
        glab, gn = ops.ctc_greedy_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda))
        assert glab[0, :int(gn[0])].cpu().tolist() == g['greedy'], name


def test_decoder_classes_reference_signature(cuda):
    """GreedyDecoder / BeamSearchDecoder called like the reference's numpy classes (probs [B,T,C])."""
    from tensorflow_end2end_speech_recognition_amd.models.ctc.decoders.greedy_decoder import GreedyDecoder
    from tensorflow_end2end_speech_recognition_amd.models.ctc.decoders.beam_search_decoder import BeamSearchDecoder
    g = np.load(os.path.join(GOLD, 'decoders_v1.npz'))
    for i in (0, 5, 8, 17):
        probs, sl = g['c%d_probs' % i], g['c%d_seq_len' % i]
        C = probs.shape[2]
        assert GreedyDecoder(blank_index=C - 1)(probs, sl)[0] == list(g['c%d_greedy' % i])
        w = int(g['c%d_widths' % i][-1])
        hyp, score = BeamSearchDecoder(space_index=-1, blank_index=C - 1)(probs, sl, beam_width=w)
        assert hyp[0] == list(g['c%d_beam%d' % (i, w)])


def test_beam_search_batch_vs_oracle(cuda):
    """A ragged batch at TIMIT shape (C=62, beam 20) and a wide-vocabulary case against the oracle."""
    ops = _ops()
    rng = np.random.RandomState(3)
    # (the last four: beams wide enough for the (entry, class) pruning of round 6 to bite -- W^2 candidates cut to
    # ~W ln W -- on smooth, peaked-with-repeats, quantised (ties inside the pruning rule's margins) and near-flat posteriors)
    for (T, B, C, W, sharp) in [(60, 6, 62, 20, 3.0), (30, 3, 300, 10, 4.0), (25, 2, 29, 100, 2.0),
                                (20, 2, 400, 8, 2.0), (24, 2, 200, 50, 2.0), (24, 2, 201, 48, 1.0), (16, 2, 600, 64, 1.5),
                                (20, 2, 150, 100, 0.05)]:
        logits = (rng.randn(T, B, C) * sharp).astype(np.float32)
        if C in (400, 201):   # coarse grid of values: many exact ties at the class-pruning threshold
            logits = np.round(logits * 2) / 2
        logits[:, :, C - 1] += sharp
        for t in range(1, T):
            if rng.rand() < 0.3:
                logits[t] = logits[t - 1]
        sl = rng.randint(1, T + 1, size=B).astype(np.int32)
        sl[0] = T
        lab, n, score = ops.ctc_beam_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda), W)
        lp = octc.log_softmax(logits.astype(np.float64).transpose(1, 0, 2))
        ref, rs = odec.beam_search_decode(lp, sl, C - 1, W)
        for b in range(B):
            assert lab[b, :int(n[b])].cpu().tolist() == ref[b], (T, C, W, b)
            assert abs(score[b].item() - rs[b]) < 1e-7 * max(1, abs(rs[b]))


@pytest.mark.parametrize('rows,E,vocab', [(12832, 64, 30), (300, 8, 12), (77, 20, 9), (500, 1024, 5)])
def test_embedding_gather_scatter(cuda, rows, E, vocab):
    """asr_embedding_gather / asr_embedding_scatter (the decoder's input embedding and its gradient) against
    indexing / index_add; the scatter is deterministic (fixed summation order)."""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(rows + E)
    W = torch.randn(vocab, E, generator=g).to(cuda)
    ids = torch.randint(0, vocab, (rows,), generator=g).to(torch.int32).to(cuda)
    dout = torch.randn(rows, E, generator=g).to(cuda)
    if rows:
        assert torch.equal(ops.embedding_gather(W, ids), W[ids.long()])
    dW = ops.embedding_scatter(dout, ids, vocab, torch.empty(vocab, E, device=cuda))
    ref = torch.zeros(vocab, E, dtype=torch.float64).index_add_(0, ids.long().cpu(), dout.double().cpu())
    assert (dW.double().cpu() - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(dW, ops.embedding_scatter(dout, ids, vocab, torch.empty(vocab, E, device=cuda)))


def test_softmax_rows(cuda):
    ops = _ops()
    x = torch.randn(100, 62, device=cuda) * 3
    assert torch.allclose(ops.softmax_rows(x), torch.softmax(x, 1), atol=1e-6)


# --------------------------------------------------------------------------- clip + optimizers
def test_clip_decay_optimizers(cuda):
    ops = _ops()
    from tensorflow_end2end_speech_recognition_amd._lib import OPTIMIZER_IDS
    rng = np.random.RandomState(0)
    sizes = [5000, 16, 12288, 48, 1]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])]).astype(np.int64)
    n = int(offs[-1])
    g = np.zeros(n, np.float32)
    p = np.zeros(n, np.float32)
    for i, s in enumerate(sizes):
        g[offs[i]:offs[i] + s] = rng.randn(s) * (10 if i % 2 == 0 else 0.001)
        p[offs[i]:offs[i] + s] = rng.randn(s)
    plan = ops.ClipPlan(offs, cuda)
    gd = torch.tensor(g, device=cuda)
    ops.clip_by_norm_multi(gd, plan, 5.0)
    ref = g.copy()
    for i, s in enumerate(sizes):
        ref[offs[i]:offs[i] + s] = oopt.clip_by_norm(g[offs[i]:offs[i] + s], 5.0)
    assert np.abs(gd.cpu().numpy() - ref).max() < 1e-5
    mask = torch.tensor([1, 0, 1, 0, 1], dtype=torch.uint8, device=cuda)
    pd = torch.tensor(p, device=cuda)
    l2 = torch.zeros((), device=cuda)
    g2 = torch.tensor(ref, device=cuda)
    ops.weight_decay(g2, pd, plan, mask, 0.01, l2_out=l2)
    exp = ref.copy()
    tot = 0.0
    for i, s in enumerate(sizes):
        if i % 2 == 0:
            exp[offs[i]:offs[i] + s] += 0.01 * p[offs[i]:offs[i] + s]
            tot += 0.5 * (p[offs[i]:offs[i] + s].astype(np.float64) ** 2).sum()
    assert np.abs(g2.cpu().numpy() - exp).max() < 1e-6
    assert abs(l2.item() - 0.01 * tot) / (0.01 * tot) < 1e-5
    for name, oid in OPTIMIZER_IDS.items():
        pr = p.astype(np.float64)
        s0, s1 = oopt.init_slots(name, pr)
        pdv = torch.tensor(p, device=cuda)
        d0 = torch.tensor(s0.astype(np.float32), device=cuda)
        d1 = torch.tensor(s1.astype(np.float32), device=cuda)
        for t in range(1, 4):
            gt = (rng.randn(n) * 0.1).astype(np.float32)
            pr, s0, s1 = oopt.step(name, pr, gt.astype(np.float64), s0, s1, 0.01, t)
            ops.optimizer_step(oid, pdv, torch.tensor(gt, device=cuda), d0, d1, 0.01, t)
        assert np.abs(pdv.cpu().numpy() - pr).max() < 2e-5, name


def test_dropout_mask(cuda):
    ops = _ops()
    m = ops.dropout_mask((1000, 257), 0.8, seed=7, offset=0, device=cuda)
    vals = torch.unique(m).cpu().tolist()
    assert len(vals) == 2 and vals[0] == 0 and abs(vals[1] - 1.25) < 1e-6
    assert abs((m > 0).float().mean().item() - 0.8) < 0.01
    m2 = ops.dropout_mask((1000, 257), 0.8, seed=7, offset=0, device=cuda)
    m3 = ops.dropout_mask((1000, 257), 0.8, seed=8, offset=0, device=cuda)
    assert torch.equal(m, m2) and not torch.equal(m, m3)


@pytest.mark.parametrize('M,N,K,dt', [(12448, 512, 2048, torch.bfloat16), (1500, 256, 128, torch.bfloat16),
                                      (100, 36, 24, torch.float32), (33, 8, 64, torch.bfloat16)])
def test_gemm_with_the_dropout_mask_formed_in_its_epilogue(cuda, M, N, K, dt):
    """asr_gemm_drop == asr_gemm_mul with the tensor asr_dropout_mask produces for the same (keep, seed, offset), bit for
    bit: the headline dx shape (lean NT kernel, mask from the Philox counter in the epilogue) and shapes that take the
    generic kernels followed by asr_dropout_apply in place."""
    ops = _ops()
    rng = np.random.RandomState(N)
    A = torch.tensor(rng.randn(M, K), dtype=torch.float32, device=cuda).to(dt)
    Bt = torch.tensor(rng.randn(N, K) * 0.1, dtype=torch.float32, device=cuda).to(dt)
    d = (0.8, 11, (4 << 32) + 3)
    mask = ops.dropout_mask((M, N), *d, cuda)
    ref = ops.gemm(A, Bt, transB=True, out_dtype=torch.float32, mul=mask)
    got = ops.gemm(A, Bt, transB=True, out_dtype=torch.float32, drop=d)
    assert torch.equal(got, ref) and float(got.abs().sum()) > 0


@pytest.mark.parametrize('N,H,W,Cin', [(3, 40, 11, 3), (70, 5, 3, 3), (2, 7, 6, 2), (1, 1, 1, 3), (700, 40, 11, 3)])
def test_direct_convolution_of_the_few_channel_first_layer(cuda, N, H, W, Cin):
    """asr_conv3x3_smallc_fwd / _bwd_weight (no patch matrix) against fp64 conv2d on the bf16-rounded operands: the
    cfg C image 40 x 11 x 3, odd sizes, more pixels than one weight-gradient slice, a single pixel, and enough pixels
    (308 k: 151 workgroup partials) for the two-pass sum of the partials."""
    ops = _ops()
    rng = np.random.RandomState(N + H * W)
    Cout = 64
    x = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.float32).to(torch.bfloat16)
    w = torch.tensor(rng.randn(3, 3, Cin, Cout) * 0.2, dtype=torch.float32).to(torch.bfloat16)
    b = torch.tensor(rng.randn(Cout) * 0.1, dtype=torch.float32)
    dpre = torch.tensor(rng.randn(N, H, W, Cout), dtype=torch.float32).to(torch.bfloat16)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(3, 2, 0, 1).clone().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, b.double(), padding=1)
    (y64 * dpre.double().permute(0, 3, 1, 2)).sum().backward()
    ref = torch.relu(y64).permute(0, 2, 3, 1)
    got = ops.conv3x3_smallc_fwd(x.to(cuda), w.to(cuda).view(9 * Cin, Cout), b.to(cuda), relu=True)
    assert got.dtype == torch.bfloat16
    assert np.abs(got.float().cpu().numpy() - ref.detach().numpy()).max() < 1e-2 * max(1.0, float(ref.abs().max()))
    dw = torch.empty(9 * Cin, Cout, device=cuda)
    ops.conv3x3_smallc_bwd_weight(x.to(cuda), dpre.to(cuda), dw)
    ref_dw = w64.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout).numpy()
    assert np.abs(dw.cpu().numpy() - ref_dw).max() < 2e-5 * max(1.0, np.abs(ref_dw).max())
    # + the bias gradient from a column of ones in the patch operand: the same weights bit for bit, bias = column sums
    dw2 = torch.full((9 * Cin, Cout), 5.0, device=cuda)
    db = torch.full((Cout,), 5.0, device=cuda)
    ops.conv3x3_smallc_bwd_weight_bias(x.to(cuda), dpre.to(cuda), dw2, db)
    assert torch.equal(dw, dw2)
    refb = dpre.double().sum(dim=(0, 1, 2)).numpy()
    assert np.abs(db.cpu().numpy() - refb).max() < 1e-5 * max(1.0, float(dpre.double().abs().sum(dim=(0, 1, 2)).max()))


@pytest.mark.parametrize('N,H,W,Cin,Cout,drop', [(3, 40, 11, 64, 64, True), (2, 20, 6, 64, 128, False), (2, 20, 6, 128, 128, True),
                                                 (70, 40, 11, 64, 64, True), (70, 20, 6, 64, 128, False), (70, 20, 6, 128, 128, True)])
def test_conv_data_gradient_with_the_relu_backward_below_in_its_epilogue(cuda, N, H, W, Cin, Cout, drop):
    """asr_conv3x3_bwd_data_relu == asr_relu_bwd(_drop)(asr_conv3x3_bwd_data(...), act_below), bit for bit."""
    ops = _ops()
    rng = np.random.RandomState(H + Cin + Cout)
    w = torch.tensor(rng.randn(3, 3, Cin, Cout) * 0.05, dtype=torch.float32, device=cuda)
    _, wb = ops.conv3x3_prep_weights(w)
    dy = torch.tensor(rng.randn(N, H, W, Cout), dtype=torch.float32, device=cuda).to(torch.bfloat16)
    act = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.float32, device=cuda).clamp_min(0).to(torch.bfloat16)
    d = (0.8, 9, (2 << 32) + 5) if drop else None
    ref = ops.relu_bwd(ops.conv3x3_bwd_data(dy, wb), act, drop=d)
    got = ops.conv3x3_bwd_data_relu(dy, wb, act, drop=d)
    assert torch.equal(got, ref) and float(got.float().abs().sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,C,drop', [(5, 40, 11, 64, True), (3, 20, 6, 128, False), (2, 5, 3, 8, True)])
def test_pool_dropout_relu_backward_in_one_pass(cuda, dtype, N, H, W, C, drop):
    """asr_maxpool2x2_relu_bwd == asr_dropout_apply -> asr_maxpool2x2_bwd -> asr_relu_bwd, bit for bit (the cfg C image
    sizes 40 x 11 x 64 and 20 x 6 x 128, and a small one with odd edges in both directions)."""
    ops = _ops()
    rng = np.random.RandomState(N * H + C)
    act = torch.tensor(rng.randn(N, H, W, C), dtype=torch.float32, device=cuda).clamp_min(0).to(dtype)
    pooled, arg = ops.maxpool2x2_fwd(act)
    dp = torch.tensor(rng.randn(*pooled.shape), dtype=torch.float32, device=cuda)
    d = (0.8, 5, (3 << 32) + 9) if drop else None
    ref = ops.relu_bwd(ops.maxpool2x2_bwd(ops.dropout_apply(dp, *d) if drop else dp, arg, H, W), act)
    got = ops.maxpool2x2_relu_bwd(dp, arg, act, drop=d)
    assert got.dtype == act.dtype and torch.equal(got, ref)
    assert float(got.float().abs().sum()) > 0


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(5, 7, 5, 64, 64), (3, 40, 11, 64, 128), (70, 6, 3, 128, 128), (2, 1, 1, 64, 64),
                                            (70, 40, 11, 64, 64), (66, 20, 6, 64, 128), (65, 20, 6, 128, 128),
                                            (300, 5, 4, 64, 64)])
def test_conv_relu_dropout_in_the_epilogue(cuda, N, H, W, Cin, Cout):
    """asr_conv3x3_fwd_drop == asr_dropout_apply(asr_conv3x3_fwd(relu)) bit for bit (tiled and image-resident kernels),
    and the backward forms that read the DROPPED activation -- asr_conv3x3_bwd_data_relu mode 2, asr_relu_bwd_scaled --
    equal the ones that re-form the Philox mask over the undropped activation."""
    ops = _ops()
    rng = np.random.RandomState(N + H + Cin + 1)
    x = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.float32, device=cuda).to(torch.bfloat16)
    w = torch.tensor(rng.randn(3, 3, Cin, Cout) * 0.05, dtype=torch.float32, device=cuda)
    b = torch.tensor(rng.randn(Cout) * 0.1, dtype=torch.float32, device=cuda)
    wf, _ = ops.conv3x3_prep_weights(w)
    d = (0.9, 21, (4 << 32) + 3)                                        # 1 / 0.9 is not exact: the scale must be formed alike
    act = ops.conv3x3_fwd(x, wf, b, relu=True)
    want = ops.dropout_apply(act, *d)
    got = ops.conv3x3_fwd_drop(x, wf, b, d)
    assert torch.equal(got, want)
    frac = float((got > 0).float().mean()) / max(float((act > 0).float().mean()), 1e-9)
    assert 0.8 < frac < 0.97                                         # ~ keep_prob of the active units survive
    # backward through this ReLU + dropout from the dropped tensor
    dout = torch.tensor(rng.randn(N, H, W, Cout), dtype=torch.float32, device=cuda)
    assert torch.equal(ops.relu_bwd_scaled(dout, got, d[0]), ops.relu_bwd(dout, act, drop=d))
    # ... and in the epilogue of the data gradient of the layer above (Cout2 -> Cout channels)
    w2 = torch.tensor(rng.randn(3, 3, Cout, 128) * 0.05, dtype=torch.float32, device=cuda)
    _, wb2 = ops.conv3x3_prep_weights(w2)
    dy = torch.tensor(rng.randn(N, H, W, 128), dtype=torch.float32, device=cuda).to(torch.bfloat16)
    ref = ops.conv3x3_bwd_data_relu(dy, wb2, act, drop=d)
    assert torch.equal(ops.conv3x3_bwd_data_relu(dy, wb2, got, drop=d, dropped=True), ref)
    assert float(ref.float().abs().sum()) > 0


@pytest.mark.parametrize('N,H,W,Cin', [(5, 40, 11, 3), (70, 40, 11, 3), (3, 7, 5, 1), (2, 1, 1, 3)])
def test_first_layer_conv_relu_dropout_in_the_epilogue(cuda, N, H, W, Cin):
    """asr_conv3x3_smallc_fwd_drop == asr_dropout_apply(asr_conv3x3_smallc_fwd(relu)), bit for bit."""
    ops = _ops()
    rng = np.random.RandomState(N + H + Cin + 2)
    x = torch.tensor(rng.randn(N, H, W, Cin), dtype=torch.float32, device=cuda).to(torch.bfloat16)
    w2d = torch.tensor(rng.randn(9 * Cin, 64) * 0.2, dtype=torch.float32, device=cuda).to(torch.bfloat16)
    b = torch.tensor(rng.randn(64) * 0.1, dtype=torch.float32, device=cuda)
    d = (0.8, 13, (1 << 32) + 6)
    want = ops.dropout_apply(ops.conv3x3_smallc_fwd(x, w2d, b, relu=True), *d)
    got = ops.conv3x3_smallc_fwd_drop(x, w2d, b, d)
    assert torch.equal(got, want) and float(got.float().abs().sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('N,H,W,C,drop', [(5, 40, 11, 64, True), (3, 20, 6, 128, True), (3, 20, 6, 128, False),
                                          (2, 5, 3, 8, True), (70, 40, 11, 64, True)])
def test_pool_dropout_in_one_pass_and_its_backward_from_the_pooled_tensor(cuda, dtype, N, H, W, C, drop):
    """asr_maxpool2x2_fwd_drop == asr_maxpool2x2_fwd + asr_dropout_apply (values and argmax), and
    asr_maxpool2x2_relu_bwd in mode 2 (mask = sign of the pooled, dropped activation; no Philox, no full-resolution
    activation read) == the mode that re-forms the mask over the activation, bit for bit."""
    ops = _ops()
    rng = np.random.RandomState(N * H + C + 3)
    act = torch.tensor(rng.randn(N, H, W, C), dtype=torch.float32, device=cuda).clamp_min(0).to(dtype)
    pooled, arg = ops.maxpool2x2_fwd(act)
    d = (0.8, 5, (3 << 32) + 9) if drop else None
    if drop:
        pd, arg2 = ops.maxpool2x2_fwd_drop(act, d)
        assert torch.equal(arg2, arg) and torch.equal(pd, ops.dropout_apply(pooled, *d))
    else:
        pd = pooled
    dp = torch.tensor(rng.randn(*pooled.shape), dtype=torch.float32, device=cuda)
    ref = ops.maxpool2x2_relu_bwd(dp, arg, act, drop=d)
    got = ops.maxpool2x2_relu_bwd(dp, arg, None, drop=d, pooled=pd, hw=(H, W))
    assert got.dtype == act.dtype and torch.equal(got, ref)
    assert float(got.float().abs().sum()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_formed_in_the_kernel_equals_the_stored_mask(cuda, dtype):
    """asr_dropout_apply == asr_dropout_mask + asr_apply_mask and asr_relu_bwd_drop == asr_relu_bwd with that mask, bit
    for bit -- whole tensors and chunks addressed by a shifted Philox block offset (how the VGG front-end's chunked
    first-layer backward uses it), sizes that are not multiples of 4 included."""
    ops = _ops()
    rng = np.random.RandomState(3)
    for n, seed, off in ((1000 * 257, 7, 0), (4099, 11, (5 << 32) + 17), (64 * 440 * 64, 3, 1 << 50)):
        x = torch.tensor(rng.randn(n), dtype=torch.float32, device=cuda).to(dtype)
        m = ops.dropout_mask((n,), 0.8, seed, off, cuda)
        assert torch.equal(ops.dropout_apply(x, 0.8, seed, off), ops.apply_mask(x, m))
        dout = torch.tensor(rng.randn(n), dtype=torch.float32, device=cuda)
        assert torch.equal(ops.relu_bwd(dout, x, drop=(0.8, seed, off)), ops.relu_bwd(dout, x, m))
        c0 = (n // 3) // 4 * 4                            # a chunk starting at a multiple of 4 elements
        assert torch.equal(ops.relu_bwd(dout[c0:].contiguous(), x[c0:].contiguous(), drop=(0.8, seed, off + c0 // 4)),
                           ops.relu_bwd(dout, x, m)[c0:])


@pytest.mark.parametrize('T,F,num_stack,num_skip,splice', [
    (37, 2, 1, 1, 11),      # VGG recipe shape: splice only (F*3 = 6 values per frame)
    (41, 3, 3, 3, 1),       # stacking only
    (29, 1, 2, 2, 5),       # both; D = 3*num_stack per channel
    (6, 2, 3, 2, 11),       # utterances shorter than the splice window
    (1, 1, 1, 1, 3),        # single frame
    (33, 4, 1, 1, 11),      # row width % 4 == 0: the LDS-table / 16-byte-store kernel
    (50, 40, 1, 1, 11),     # the VGG recipe's real frame: 40 channels x 3, splice 11
    (29, 4, 2, 2, 5),       # vector kernel with stacked frames
    (5, 4, 3, 3, 11),       # vector kernel, utterances shorter than the window, slots that stay zero
])
def test_device_batch_assembly(cuda, T, F, num_stack, num_skip, splice):
    """asr_stack_frames + asr_splice through utils/io/inputs/device.py assemble(): bit-exact against the host
    path of DatasetBase (stack_frame + do_splice per utterance, both pinned to the reference's outputs by
    tests/golden/splice_v1.npz), ragged lengths, zero padding."""
    from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.device import assemble
    from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.frame_stacking import stack_frame
    from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.splicing import do_splice
    rng = np.random.RandomState(T * 7 + splice)
    B, D = 5, F * 3
    lens = [T, max(1, T - 1), max(1, T // 2), max(1, T // 3), 1]
    raw = np.zeros((B, T, D), dtype=np.float32)
    for b in range(B):
        raw[b, :lens[b]] = rng.randn(lens[b], D)
    stacking = num_stack != 1
    want_rows = []
    for b in range(B):
        u = raw[b, :lens[b]].astype(np.float64)
        if stacking:
            u = np.asarray(stack_frame([u], num_stack, num_skip)[0], dtype=np.float64)
        u = do_splice(u[None], splice=splice, batch_size=1, num_stack=num_stack if stacking else 1)[0]
        want_rows.append(u.astype(np.float32))
    x, sl = assemble(raw, np.asarray(lens, np.int32), num_stack if stacking else None, num_skip if stacking else None,
                     splice, device=cuda)
    x, sl = x.cpu().numpy(), sl.cpu().numpy()
    Tn = -(-T // num_skip) if stacking else T
    assert x.shape == (B, Tn, D * (num_stack if stacking else 1) * splice)
    for b in range(B):
        n = want_rows[b].shape[0]
        assert sl[b] == n
        assert np.array_equal(x[b, :n], want_rows[b]), b
        assert not x[b, n:].any()
    with pytest.raises(ValueError):
        _ops().stack_frames(torch.zeros((1, 4, 3), device=cuda), torch.ones(1, dtype=torch.int32, device=cuda), 2, 3)


@pytest.mark.parametrize('M,N,dtype', [(300007, 64, 'bf16'), (70001, 128, 'bf16'), (5000, 256, 'bf16'), (4100, 24, 'bf16'),
                                       (1000, 64, 'bf16'), (33000, 62, 'f32')])
def test_colsum_paths(cuda, M, N, dtype):
    """asr_colsum (bias gradients: models/ctc/ctc.py output FC, the VGG stack's convolutions): the whole-row vector
    kernel for tall bf16 inputs (N % 8 == 0) and the generic two-stage kernels, against an fp64 sum of the same
    (rounded) values; deterministic (two runs bit-identical)."""
    ops = _ops()
    rng = np.random.RandomState(M % 1000 + N)
    a = torch.tensor(rng.randn(M, N).astype(np.float32), device=cuda)
    if dtype == 'bf16':
        a = a.to(torch.bfloat16)
    want = a.double().sum(0).cpu().numpy()
    got = ops.colsum(a).cpu().numpy()
    got2 = ops.colsum(a).cpu().numpy()
    assert np.array_equal(got, got2)
    assert np.abs(got - want).max() < 2e-5 * a.double().abs().sum(0).max().item()


def test_upload_ints_is_one_copy_off_the_main_stream(cuda):
    """ops.upload_ints: several host vectors arrive through one staging buffer (each on a 16-byte boundary), device
    vectors pass through, and the current stream -- and a side lane forked from it -- sees the values."""
    ops = _ops()
    rng = np.random.RandomState(5)
    busy = torch.randn(4096, 4096, device=cuda)
    for rep in range(6):                                     # the staging block / device block get reused across calls
        (busy @ busy).sum()                                  # main stream busy: the upload must not wait behind it
        lens = rng.randint(1, 900, size=64).astype(np.int64)
        offs = np.cumsum(rng.randint(0, 50, size=65)).astype(np.int32)
        flat = rng.randint(0, 28, size=int(offs[-1]) + rep).astype(np.int32)
        on_dev = torch.arange(7, device=cuda, dtype=torch.int64)
        a, b, c, d, e = ops.upload_ints(cuda, [lens, offs, flat, on_dev, np.zeros(0, np.int32)])
        assert all(t.dtype == torch.int32 and t.is_cuda for t in (a, b, c, d, e)) and e.numel() == 0
        assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and c.data_ptr() % 16 == 0
        with ops.side_lane(cuda, lane=1):
            on_lane = (a.long().sum() + c.long().sum()).clone()
        ops.join_side(cuda)
        assert int(on_lane) == int(lens.sum() + flat.sum())
        assert np.array_equal(a.cpu().numpy(), lens) and np.array_equal(b.cpu().numpy(), offs)
        assert np.array_equal(c.cpu().numpy(), flat) and np.array_equal(d.cpu().numpy(), np.arange(7))


def test_beam_search_thread_counts_agree(cuda, monkeypatch):
    """asr_ctc_beam_decode with 512 threads per utterance (default) against the four-wave form (ASR_BEAM_THREADS=256):
    same labels and lengths on flat, peaked and tie-heavy posteriors (quantised logits: many equal candidate totals, so
    the tie-index passes run), small and large vocabularies, ragged lengths; scores to 1e-9 (the fp64 log-softmax sums
    in a different order)."""
    ops = _ops()
    rng = np.random.RandomState(11)
    for T, B, C, W, kind in ((60, 5, 62, 20, 'flat'), (60, 5, 62, 20, 'ties'), (40, 3, 3387, 100, 'peaked'),
                             (40, 3, 3387, 100, 'ties'), (50, 4, 1000, 40, 'flat'), (30, 2, 30, 128, 'flat'),
                             (25, 2, 5000, 64, 'peaked')):
        if kind == 'flat':
            lg = rng.randn(T, B, C).astype(np.float32) * 3
        elif kind == 'ties':
            lg = np.round(rng.randn(T, B, C).astype(np.float32) * 2) / 2          # steps of 0.5: equal totals abound
        else:
            lg = rng.randn(T, B, C).astype(np.float32)
            win = np.where(rng.rand(T, B) < 0.6, C - 1, rng.randint(0, C - 1, size=(T, B)))
            np.put_along_axis(lg, win[:, :, None], 12.0 + rng.rand(T, B, 1).astype(np.float32), axis=2)
        logits = torch.tensor(lg, device=cuda)
        sl = torch.tensor(rng.randint(T // 2, T + 1, size=B).astype(np.int32), device=cuda)
        out = {}
        for nt in ('256', '512'):
            monkeypatch.setenv('ASR_BEAM_THREADS', nt)
            lab, n, score = ops.ctc_beam_decode(logits, sl, beam_width=W)
            out[nt] = (lab.cpu().numpy(), n.cpu().numpy(), score.cpu().numpy())
        monkeypatch.delenv('ASR_BEAM_THREADS')
        assert np.array_equal(out['256'][1], out['512'][1]), (T, B, C, W, kind)
        assert np.array_equal(out['256'][0], out['512'][0]), (T, B, C, W, kind)
        assert np.abs(out['256'][2] - out['512'][2]).max() < 1e-9 * max(1.0, np.abs(out['256'][2]).max())


def test_beam_search_edge_inputs_against_the_oracle(cuda):
    """Prefix beam search on inputs at the edges of its bookkeeping: width 1 and the maximum width 128, two classes,
    constant logits (every candidate of a frame ties: the insertion order alone decides), classes masked with -inf, an empty
    and a one-frame utterance in the batch -- labels identical to oracle.decoders.beam_search_decode."""
    ops = _ops()
    rng = np.random.RandomState(21)
    cases = []
    for T, B, C, W in ((12, 3, 2, 1), (10, 2, 5, 128), (9, 2, 40, 128), (14, 3, 30, 7), (8, 2, 700, 100)):
        cases.append((rng.randn(T, B, C).astype(np.float32) * 2, W, 'random'))
    cases.append((np.zeros((10, 2, 12), np.float32), 16, 'constant'))
    cases.append((np.zeros((6, 2, 300), np.float32), 64, 'constant wide'))
    masked = rng.randn(12, 3, 50).astype(np.float32)
    masked[:, :, 5:30] = -np.inf
    cases.append((masked, 20, 'masked'))
    for logits, W, kind in cases:
        T, B, C = logits.shape
        sl = np.full(B, T, np.int32)
        sl[-1] = 1
        if B > 2:
            sl[1] = 0
        lab, n, score = ops.ctc_beam_decode(torch.tensor(logits, device=cuda), torch.tensor(sl, device=cuda), W)
        lp = octc.log_softmax(logits.astype(np.float64).transpose(1, 0, 2))
        ref, rs = odec.beam_search_decode(lp, sl, C - 1, W)
        for b in range(B):
            assert lab[b, :int(n[b])].cpu().tolist() == ref[b], (kind, T, C, W, b)
            if sl[b] > 0:
                assert abs(score[b].item() - rs[b]) < 1e-7 * max(1, abs(rs[b])), (kind, b)


@pytest.mark.parametrize('H,B,T,ndir', [(128, 16, 40, 2), (256, 16, 40, 2), (256, 32, 33, 1), (128, 48, 21, 2), (64, 32, 27, 2), (320, 16, 23, 2)])
def test_gru_cluster_forward_matches_the_single_cu_kernel(cuda, monkeypatch, H, B, T, ndir):
    """asr_gru_fwd on clusters of H / 32 CUs (three-bf16-term products, two exchanges per step) against the single-CU
    persistent kernel (exact-fp32 MFMA; ASR_GRU_CLUSTER=0): r, u, c, r * h at every frame a row worked on, hout everywhere
    (zero past a row's length), the final state; ragged lengths incl. an empty row; no hand-off flag raised."""
    ops = _ops()
    rng = np.random.RandomState(H + B + T)
    xg = torch.tensor(rng.randn(T, B, ndir * 2 * H) * 0.5, dtype=torch.float32, device=cuda)
    xc = torch.tensor(rng.randn(T, B, ndir * H) * 0.5, dtype=torch.float32, device=cuda)
    wgh = torch.tensor(rng.randn(ndir, H, 2 * H) * 0.08, dtype=torch.float32, device=cuda)
    wch = torch.tensor(rng.randn(ndir, H, H) * 0.08, dtype=torch.float32, device=cuda)
    sl_np = rng.randint(1, T + 1, size=B).astype(np.int32)
    sl_np[0] = T
    if B > 16:
        sl_np[17] = 0
    sl = torch.tensor(sl_np, device=cuda)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('ASR_GRU_CLUSTER', mode)
        out[mode] = {k: v.cpu().numpy() for k, v in ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir).items()}
    monkeypatch.delenv('ASR_GRU_CLUSTER')
    assert ops.check_async_errors(0) == 0
    valid = (np.arange(T)[:, None] < sl_np[None, :])[:, :, None]                  # [T,B,1]
    for k in ('r', 'u', 'c', 'rh', 'hout'):                                       # (r, u, c past a row's length: unspecified)
        a, b = out['0'][k], out['1'][k]
        assert np.abs(np.where(valid, a - b, 0.0)).max() < 3e-6, k
        assert np.isfinite(np.where(valid, b, 0.0)).all(), k
    assert np.abs(np.where(valid, 0.0, out['1']['hout'])).max() == 0.0
    assert np.abs(np.where(valid, 0.0, out['1']['rh'])).max() == 0.0             # the caller's zeros are left alone
    assert np.abs(out['0']['h_final'] - out['1']['h_final']).max() < 3e-6


@pytest.mark.parametrize('H,B,T,ndir', [(128, 16, 40, 2), (256, 16, 40, 2), (256, 32, 33, 1), (128, 48, 21, 2), (64, 32, 27, 2), (320, 16, 23, 2)])
def test_gru_cluster_backward_matches_the_single_cu_kernel(cuda, monkeypatch, H, B, T, ndir):
    """asr_gru_bwd on clusters (two all-gathers per step: dc_pre + du_pre, then dr_pre; W^T columns of the CU's units in
    registers) against the single-CU persistent kernel on the same saved activations: dgate / dcand within 2e-6 (of values up
    to ~1), exact zeros wherever no row is active, a final-state gradient fed in, ragged lengths incl. an empty row."""
    ops = _ops()
    rng = np.random.RandomState(2 * H + B + T)
    xg = torch.tensor(rng.randn(T, B, ndir * 2 * H) * 0.5, dtype=torch.float32, device=cuda)
    xc = torch.tensor(rng.randn(T, B, ndir * H) * 0.5, dtype=torch.float32, device=cuda)
    wgh = torch.tensor(rng.randn(ndir, H, 2 * H) * 0.08, dtype=torch.float32, device=cuda)
    wch = torch.tensor(rng.randn(ndir, H, H) * 0.08, dtype=torch.float32, device=cuda)
    sl_np = rng.randint(1, T + 1, size=B).astype(np.int32)
    sl_np[0] = T
    if B > 16:
        sl_np[17] = 0
    sl = torch.tensor(sl_np, device=cuda)
    monkeypatch.setenv('ASR_GRU_CLUSTER', '0')
    saved = ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)
    dout = torch.tensor(rng.randn(T, B, ndir * H) * 0.3, dtype=torch.float32, device=cuda)
    dhf = torch.tensor(rng.randn(ndir, B, H) * 0.3, dtype=torch.float32, device=cuda)
    wghT, wchT = wgh.transpose(1, 2).contiguous(), wch.transpose(1, 2).contiguous()
    got = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('ASR_GRU_CLUSTER', mode)
        dg, dc = ops.gru_bwd(dout, dhf, saved, wghT, wchT, sl, T, H, ndir)
        got[mode] = (dg.cpu().numpy(), dc.cpu().numpy())
    monkeypatch.delenv('ASR_GRU_CLUSTER')
    assert ops.check_async_errors(0) == 0
    for a, b in zip(got['0'], got['1']):
        assert np.isfinite(b).all() and np.abs(a - b).max() < 2e-6 and np.abs(a).max() > 0.1
    valid = (np.arange(T)[:, None] < sl_np[None, :])[:, :, None]
    assert np.abs(np.where(valid, 0.0, got['1'][0])).max() == 0.0 and np.abs(np.where(valid, 0.0, got['1'][1])).max() == 0.0


def test_gru_cluster_handoff_timeout_is_reported(cuda):
    """The GRU clusters use the LSTM clusters' bounded polls and sticky error word: with the test-only flag (one member of
    every cluster leaves early, spin limit 2000 polls) a forward and a backward launch finish, and the blocking check
    raises; the word is reported once and cleared."""
    from tensorflow_end2end_speech_recognition_amd import _lib
    if os.environ.get('ASR_GRU_CLUSTER') == '0' or os.environ.get('ASR_GRU_PERSISTENT') == '0':
        pytest.skip('the GRU clusters are switched off in this run')
    ops = _ops()
    rng = np.random.RandomState(4)
    H, B, T, ndir = 128, 16, 9, 2
    xg = torch.tensor(rng.randn(T, B, ndir * 2 * H) * 0.5, dtype=torch.float32, device=cuda)
    xc = torch.tensor(rng.randn(T, B, ndir * H) * 0.5, dtype=torch.float32, device=cuda)
    wgh = torch.tensor(rng.randn(ndir, H, 2 * H) * 0.08, dtype=torch.float32, device=cuda)
    wch = torch.tensor(rng.randn(ndir, H, H) * 0.08, dtype=torch.float32, device=cuda)
    sl = torch.full((B,), T, dtype=torch.int32, device=cuda)
    assert ops.check_async_errors(0) == 0
    saved = ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)
    try:
        ops.debug_set_lstm_flags(64)
        ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)
        with pytest.raises(_lib.AsrError):
            ops.check_async_errors(0)
        assert ops.check_async_errors(0) == 0
        dout = torch.zeros((T, B, ndir * H), dtype=torch.float32, device=cuda)
        ops.gru_bwd(dout, None, saved, wgh.transpose(1, 2).contiguous(), wch.transpose(1, 2).contiguous(), sl, T, H, ndir)
        with pytest.raises(_lib.AsrError):
            ops.check_async_errors(0)
    finally:
        ops.debug_set_lstm_flags(0)
    assert ops.check_async_errors(0) == 0
    out = ops.gru_fwd(xg, xc, wgh, wch, sl, T, H, ndir)             # and the clusters work again
    assert float((out['hout'] - saved['hout']).abs().max()) == 0.0 and ops.check_async_errors(0) == 0
