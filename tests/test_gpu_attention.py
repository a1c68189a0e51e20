"""GPU parity: attention encoder-decoder and joint CTC-attention (class surface of
models/attention/*.py) vs the oracle -- loss within 1e-4 relative (fp32), every parameter gradient,
teacher-forced logits, greedy inference ids bit-exact."""
import numpy as np
import pytest
import torch

from oracle import attention as oatt

pytestmark = pytest.mark.gpu


def _batch(rng, B, T, D, C):
    x = rng.randn(B, T, D).astype(np.float32)
    sl = rng.randint(max(2, T // 2), T + 1, size=B).astype(np.int32)
    sl[0] = T
    lens = rng.randint(1, 6, size=B)
    lens[0] = 5
    Lmax = int(lens.max()) + 2
    sos, eos = C, C + 1
    labels = np.full((B, Lmax), eos, dtype=np.int64)
    ctc_labels = np.full((B, int(lens.max())), -1, dtype=np.int64)
    for b in range(B):
        x[b, sl[b]:] = 0
        y = rng.randint(0, C, size=lens[b])
        labels[b, 0] = sos
        labels[b, 1:1 + lens[b]] = y
        ctc_labels[b, :lens[b]] = y
    return x, sl, labels, lens + 2, ctc_labels


def _mk(cls, att, D, H, L, U, A, Em, C, **kw):
    return cls(input_size=D, encoder_type='blstm', encoder_num_units=H, encoder_num_layers=L,
               encoder_num_proj=None, attention_type=att, attention_dim=A, decoder_type='lstm',
               decoder_num_units=U, decoder_num_layers=1, embedding_dim=Em, num_classes=C, sos_index=C,
               eos_index=C + 1, max_decode_length=12, parameter_init=0.1, clip_grad_norm=5.0,
               clip_activation_encoder=50, clip_activation_decoder=50, dtype='f32', seed=5, **kw)


@pytest.mark.parametrize('att', ['bahdanau_content', 'location', 'hybrid', 'dot_product', 'luong_dot', 'luong_general',
                                 'luong_concat'])
def test_attention_model_parity(cuda, att):
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    rng = np.random.RandomState(11)
    B, T, D, H, L, U, A, Em, C = 5, 17, 12, 64, 1, 128, 32, 8, 9
    x, sl, labels, lsl, _ = _batch(rng, B, T, D, C)
    model = _mk(AttentionSeq2Seq, att, D, H, L, U, A, Em, C, sharpening_factor=1.5, logits_temperature=2.0)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, att, clip_enc=50.0, clip_dec=50.0,
                                       sharpening=1.5, temperature=2.0)
    loss, logits, out_train, out_infer = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(logits.cpu().numpy() * 1.0 - ref['logits'] * 2.0).max() < 2e-4      # ref logits are /temperature
    assert np.abs(out_train.attention_weights.cpu().numpy() - ref['alphas']).max() < 1e-5
    assert np.array_equal(out_train.predicted_ids.cpu().numpy(), ref['predicted_ids'])
    opt = model._set_optimizer('adam', 1e-3)
    gv = opt.compute_gradients(loss, model=model)
    for g, name in gv:
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
    # greedy inference
    ids = out_infer.predicted_ids.cpu().numpy()
    ref_ids = oatt.attention_model_infer(sd, x, sl, L, att, C, C + 1, 12, clip_enc=50.0, clip_dec=50.0, sharpening=1.5)
    assert np.array_equal(ids, ref_ids)
    dtr, dinf = model.decode(out_train, out_infer)
    assert dtr.shape[0] == B and dinf.shape[0] == B


@pytest.mark.parametrize('att', ['bahdanau_content', 'luong_dot'])
def test_attention_sigmoid_smoothing(cuda, att):
    """sigmoid_smoothing=True (attention_layer.py:92-96): alpha = sigmoid(e) / sum sigmoid(e); loss, weights,
    every gradient and the greedy decode against the fp64 oracle."""
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    rng = np.random.RandomState(21)
    B, T, D, H, L, U, A, Em, C = 5, 19, 12, 64, 1, 128, 32, 8, 9
    x, sl, labels, lsl, _ = _batch(rng, B, T, D, C)
    model = _mk(AttentionSeq2Seq, att, D, H, L, U, A, Em, C, sharpening_factor=2.0, sigmoid_smoothing=True)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, att, clip_enc=50.0, clip_dec=50.0, sharpening=2.0,
                                       sigmoid_smoothing=True)
    loss, logits, out_train, out_infer = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    al = out_train.attention_weights.cpu().numpy()
    assert np.abs(al - ref['alphas']).max() < 1e-5
    opt = model._set_optimizer('adam', 1e-3)
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
    ref_ids = oatt.attention_model_infer(sd, x, sl, L, att, C, C + 1, 12, clip_enc=50.0, clip_dec=50.0, sharpening=2.0,
                                         sigmoid_smoothing=True)
    assert np.array_equal(out_infer.predicted_ids.cpu().numpy(), ref_ids)


def test_joint_ctc_attention_parity_and_training(cuda):
    from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    rng = np.random.RandomState(3)
    B, T, D, H, L, U, A, Em, C = 6, 20, 12, 64, 2, 64, 32, 8, 7
    x, sl, labels, lsl, ctc_labels = _batch(rng, B, T, D, C)
    model = _mk(JointCTCAttention, 'bahdanau_content', D, H, L, U, A, Em, C, lambda_weight=0.5)
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ctc_list = [[int(v) for v in row if v >= 0] for row in ctc_labels]
    ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, 'bahdanau_content', clip_enc=50.0, clip_dec=50.0,
                                       ctc_labels=ctc_list, lambda_weight=0.5)
    loss, logits, ctc_logits, otr, oinf = model.compute_loss(x, labels, list2sparsetensor(ctc_labels, -1), sl, lsl,
                                                             1.0, 1.0, 1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(ctc_logits.cpu().numpy() - ref['ctc_logits']).max() < 1e-4
    assert np.abs(model.ctc_losses.cpu().numpy() - ref['ctc_losses']).max() / ref['ctc_losses'].max() < 1e-4
    opt = model._set_optimizer('adam', 1e-3)
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err)
    # overfit the batch (the reference's own test strategy) incl. dropout paths
    first = last = None
    for step in range(40):
        loss, *_ = model.compute_loss(x, labels, ctc_labels, sl, lsl, 0.9, 0.9, 0.9)
        model.train(loss, 'adam', 2e-3)
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < 0.6 * first, (first, last)


def test_attention_bf16_operands(cuda):
    """bf16-operand model (encoder MFMA operands, per-step context / d-alpha streams over the bf16 encoder
    copy): loss within bf16 rounding of the fp64 oracle, and it trains."""
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    rng = np.random.RandomState(8)
    B, T, D, H, L, U, A, Em, C = 6, 20, 12, 64, 2, 64, 32, 8, 7
    x, sl, labels, lsl, _ = _batch(rng, B, T, D, C)
    for att in ('bahdanau_content', 'luong_dot'):
        kw = dict(input_size=D, encoder_type='blstm', encoder_num_units=H, encoder_num_layers=L, encoder_num_proj=None,
                  attention_type=att, attention_dim=A, decoder_type='lstm', decoder_num_units=2 * H if att == 'luong_dot' else U,
                  decoder_num_layers=1, embedding_dim=Em, num_classes=C, sos_index=C, eos_index=C + 1,
                  max_decode_length=12, parameter_init=0.1, clip_grad_norm=5.0, clip_activation_encoder=50,
                  clip_activation_decoder=50, seed=5)
        model = AttentionSeq2Seq(dtype='bf16', **kw)
        sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
        ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, att, clip_enc=50.0, clip_dec=50.0)
        loss, *_ = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
        assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 3e-2, att
        first = None
        for step in range(30):
            loss, *_ = model.compute_loss(x, labels, sl, lsl, 0.9, 0.9, 0.9)
            model.train(loss, 'adam', 3e-3)
            first = loss.item() if first is None else first
        assert loss.item() < 0.7 * first, att


# ------------------------------------------------------------------------------------------------------------------
# Kernel-level parity at cfg-D / cfg-E sequence lengths (T up to 1600, A = 128, 2H = 1024): the scoring /
# softmax / context kernels work on 64-frame chunks with per-chunk partials reduced in a fixed order, so every
# T > 64 exercises the multi-chunk path the small model tests never reach.
def _ragged(rng, B, T):
    sl = rng.randint(max(1, T // 3), T + 1, size=B).astype(np.int32)
    sl[0] = T
    if B > 2:
        sl[2] = min(T, 3)
    return sl


@pytest.mark.parametrize('T', [65, 200, 1600])
@pytest.mark.parametrize('mode,with_keys', [(0, True), (0, False), (1, True)])
def test_att_energy_kernels_multi_chunk(cuda, T, mode, with_keys):
    """asr_att_energy_fwd / _bwd (attention_layer.py:162-186, 262): additive with keys (bahdanau / hybrid), additive
    without keys (location, Q6) and dot-product energies, vs fp64 autograd."""
    from tensorflow_end2end_speech_recognition_amd import ops
    rng = np.random.RandomState(T + mode)
    B, A = 3, 128
    keys = torch.tensor(rng.randn(T, B, A) * 0.5, dtype=torch.float32) if with_keys else None
    qz = torch.tensor(rng.randn(B, A) * 0.5, dtype=torch.float32)
    v = torch.tensor(rng.randn(A), dtype=torch.float32)
    de = torch.tensor(rng.randn(B, T), dtype=torch.float32)
    k64 = keys.double().clone().requires_grad_(True) if with_keys else None
    q64, v64 = qz.double().clone().requires_grad_(True), v.double().clone().requires_grad_(True)
    if mode == 0:
        z = q64.unsqueeze(0) + (k64 if with_keys else torch.zeros(T, B, A, dtype=torch.float64))
        ref = (v64 * torch.tanh(z)).sum(2).t()
    else:
        ref = (k64 * q64.unsqueeze(0)).sum(2).t()
    (ref * de.double()).sum().backward()
    kd = keys.to(cuda) if with_keys else None
    got = ops.att_energy_fwd(kd, qz.to(cuda), v.to(cuda) if mode == 0 else None, T, mode)
    assert np.abs(got.cpu().numpy() - ref.detach().numpy()).max() < 2e-5 * max(1.0, float(ref.abs().max()))
    dkeys = torch.zeros(T, B, A, device=cuda) if with_keys else None
    dqz, dv = ops.att_energy_bwd(de.to(cuda), kd, qz.to(cuda), v.to(cuda) if mode == 0 else None, mode, dkeys=dkeys,
                                 want_dv=mode == 0)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(dqz.cpu().numpy(), q64.grad.numpy()) < 1e-5
    if mode == 0:
        assert rel(dv.sum(0).cpu().numpy(), v64.grad.numpy()) < 1e-5
    if with_keys:
        assert rel(dkeys.cpu().numpy(), k64.grad.numpy()) < 1e-5


@pytest.mark.parametrize('T', [65, 200, 1600])
@pytest.mark.parametrize('enc_dtype,sigmoid', [('f32', False), ('bf16', False), ('f32', True)])
def test_att_softmax_context_kernels_multi_chunk(cuda, T, enc_dtype, sigmoid):
    """asr_att_softmax_ctx_fwd / _bwd (attention_layer.py:75-111): mask, sharpening, softmax / sigmoid smoothing,
    context, and the gradients w.r.t. the energies and (per-step form) the encoder outputs, incl. an external
    gradient w.r.t. alpha (carried location features), vs fp64 autograd on the operands the kernel reads."""
    from tensorflow_end2end_speech_recognition_amd import ops
    rng = np.random.RandomState(T + 7)
    B, E, sharp = 4, 1024, 1.5
    sl = _ragged(rng, B, T)
    enc = torch.tensor(rng.randn(T, B, E), dtype=torch.float32)
    if enc_dtype == 'bf16':
        enc = enc.to(torch.bfloat16)
    energy = torch.tensor(rng.randn(B, T) * 2, dtype=torch.float32)
    dctx = torch.tensor(rng.randn(B, E), dtype=torch.float32)
    dextra = torch.tensor(rng.randn(B, T), dtype=torch.float32)
    e64 = energy.double().clone().requires_grad_(True)
    enc64 = enc.double().clone().requires_grad_(True)
    mask = torch.arange(T).unsqueeze(0) < torch.tensor(sl).long().unsqueeze(1)
    em = torch.where(mask, e64, torch.full_like(e64, float(np.finfo(np.float32).min))) * sharp
    if sigmoid:
        sg = torch.sigmoid(em) * mask
        alpha = sg / sg.sum(1, keepdim=True)
    else:
        alpha = torch.softmax(em, 1)
    ctx = torch.einsum('bt,tbe->be', alpha, enc64)
    ((ctx * dctx.double()).sum() + (alpha * dextra.double() * mask).sum()).backward()
    sld = torch.tensor(sl, device=cuda)
    snorm = torch.empty(B, device=cuda) if sigmoid else None
    a_dev, c_dev = ops.att_softmax_ctx_fwd(energy.to(cuda), sld, sharp, enc.to(cuda), sigmoid_norm=snorm)
    assert np.abs(a_dev.cpu().numpy() - alpha.detach().numpy()).max() < 2e-6
    assert np.abs(c_dev.cpu().numpy() - ctx.detach().numpy()).max() < 2e-5 * max(1.0, float(ctx.abs().max()))
    denc = torch.zeros(T, B, E, device=cuda)
    de_dev = ops.att_softmax_ctx_bwd(dctx.to(cuda), a_dev, sld, sharp, enc.to(cuda), denc, sigmoid_norm=snorm,
                                     dalpha_extra=dextra.to(cuda))
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(de_dev.cpu().numpy(), e64.grad.numpy()) < 2e-5
    assert rel(denc.cpu().numpy(), enc64.grad.numpy()) < 2e-5


@pytest.mark.parametrize('T', [30, 65, 200, 1600])
@pytest.mark.parametrize('taps,with_keys', [(201, False), (200, True)])
def test_att_location_feature_kernels(cuda, T, taps, with_keys):
    """asr_att_loc_energy_fwd / _bwd: conv1d(alpha_prev, filter [taps,1,10], SAME) -> W_filter inside the energy
    (attention_layer.py:200-229 hybrid, 200 taps, pad 99 / 100; :239-265 location, 201 taps), two consecutive
    steps' accumulation of the per-utterance filter / W_filter gradients, vs fp64 autograd of the torch statement."""
    from tensorflow_end2end_speech_recognition_amd import ops
    from _cpu_ops import _loc_energy
    rng = np.random.RandomState(T + taps)
    B, A = 3, 128
    keys = torch.tensor(rng.randn(T, B, A) * 0.5, dtype=torch.float32) if with_keys else None
    qz = torch.tensor(rng.randn(B, A) * 0.5, dtype=torch.float32)
    v = torch.tensor(rng.randn(A), dtype=torch.float32)
    filt = torch.tensor(rng.randn(taps, 1, 10) * 0.3, dtype=torch.float32)
    wfil = torch.tensor(rng.randn(10, A) * 0.3, dtype=torch.float32)
    sl = _ragged(rng, B, T)
    ap = torch.tensor(rng.rand(B, T), dtype=torch.float32)
    ap = ap * (torch.arange(T).unsqueeze(0) < torch.tensor(sl).long().unsqueeze(1))
    ap = ap / ap.sum(1, keepdim=True)                       # a softmax-like previous weight vector, zero past len
    de = [torch.tensor(rng.randn(B, T), dtype=torch.float32) for _ in range(2)]
    leaves = [t.double().clone().requires_grad_(True) for t in (ap, filt, wfil, qz, v)]
    k64 = keys.double().clone().requires_grad_(True) if with_keys else None
    ref = _loc_energy(leaves[0], leaves[1], leaves[2], k64, leaves[3], leaves[4], T)
    dev = lambda t: None if t is None else t.to(cuda)
    got = ops.att_loc_energy_fwd(dev(ap), dev(filt), dev(wfil), dev(keys), dev(qz), dev(v), T)
    assert np.abs(got.cpu().numpy() - ref.detach().numpy()).max() < 3e-5 * max(1.0, float(ref.abs().max()))
    (ref * (de[0] + de[1]).double()).sum().backward()       # two "steps" with the same operands: grads add up
    dw = torch.empty(B, 10, A, device=cuda)
    df = torch.empty(B, taps, 10, device=cuda)
    dkeys = torch.zeros(T, B, A, device=cuda) if with_keys else None
    dq = dv = dap = 0
    for i in range(2):
        a, b_, c = ops.att_loc_energy_bwd(dev(de[i]), dev(ap), dev(filt), dev(wfil), dev(keys), dev(qz), dev(v), dw, df,
                                          accumulate=(i == 1), dkeys=dkeys)
        dq, dv, dap = dq + a, dv + b_, dap + c
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(dq.cpu().numpy(), leaves[3].grad.numpy()) < 2e-5
    assert rel(dv.sum(0).cpu().numpy(), leaves[4].grad.numpy()) < 2e-5
    assert rel(dw.sum(0).cpu().numpy(), leaves[2].grad.numpy()) < 2e-5
    assert rel(df.sum(0).cpu().numpy().reshape(taps, 1, 10), leaves[1].grad.numpy()) < 2e-5
    assert rel(dap.cpu().numpy(), leaves[0].grad.numpy()) < 2e-5
    if with_keys:
        assert rel(dkeys.cpu().numpy(), k64.grad.numpy()) < 2e-5


@pytest.mark.parametrize('att,sig,B,T', [('location', False, 5, 17), ('hybrid', False, 5, 17), ('hybrid', True, 4, 150),
                                         ('location', False, 4, 150), ('bahdanau_content', False, 4, 150)])
def test_attention_model_parity_carried_alpha_and_long_inputs(cuda, att, sig, B, T):
    """prev_alpha='carry' (the recurrence attention_layer.py:191-265 was written to express):
    loss, weights, EVERY gradient (incl. `filter`, W_filter/weights, which are dead under the reference's effective
    graph) and greedy inference vs the oracle; T = 150 spans three 64-frame chunks of the scoring kernels and the
    per-utterance alpha^T dctx contraction of the encoder gradient."""
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    rng = np.random.RandomState(31 + T)
    D, H, L, U, A, Em, C = 12, 64, 1, 128, 32, 8, 9
    x, sl, labels, lsl, _ = _batch(rng, B, T, D, C)
    model = _mk(AttentionSeq2Seq, att, D, H, L, U, A, Em, C, sharpening_factor=1.5, sigmoid_smoothing=sig,
                prev_alpha='carry')
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ref = oatt.attention_model_forward(sd, x, labels, sl, lsl, L, att, clip_enc=50.0, clip_dec=50.0, sharpening=1.5,
                                       sigmoid_smoothing=sig, prev_alpha='carry')
    loss, logits, out_train, out_infer = model.compute_loss(x, labels, sl, lsl, 1.0, 1.0, 1.0)
    assert abs(loss.item() - ref['total_loss']) / abs(ref['total_loss']) < 1e-4
    assert np.abs(out_train.attention_weights.cpu().numpy() - ref['alphas']).max() < 1e-5
    opt = model._set_optimizer('adam', 1e-3)
    A_ = 'attention_decoder/decoder/attention_layer/'
    for g, name in opt.compute_gradients(loss, model=model):
        r = ref['grads'][name]
        err = np.abs(g.cpu().numpy() - r).max()
        assert err < 2e-3 * max(np.abs(r).max(), 1e-3) + 1e-7, (name, err, np.abs(r).max())
        if att != 'bahdanau_content' and name in (A_ + 'filter', A_ + 'W_filter/weights'):
            assert np.abs(r).max() > 0          # the location path is live
    ref_ids = oatt.attention_model_infer(sd, x, sl, L, att, C, C + 1, 12, clip_enc=50.0, clip_dec=50.0, sharpening=1.5,
                                         sigmoid_smoothing=sig, prev_alpha='carry')
    assert np.array_equal(out_infer.predicted_ids.cpu().numpy(), ref_ids)


@pytest.mark.parametrize('B,K,U,peep,bias,mask', [(32, 1600, 512, True, True, True), (5, 192, 64, False, True, False),
                                                  (17, 640, 128, True, False, True), (1, 64, 8, False, False, False)])
def test_cell_product_and_cell_in_one_launch(cuda, B, K, U, peep, bias, mask):
    """asr_lstm_cell_gemm_fwd on the gate-interleaved image (asr_lstm_cell_gemm_prep) == asr_gemm_act -> asr_lstm_cell_fwd_ex,
    bit for bit: every output, the copies into column blocks of wider arrays included; finished rows (live = 0) copy their
    state through; the cfg D decoder's own shape (B = 32, 1600 -> 4 x 512) and ragged ones.  Reference:
    attention_decoder.py:142-229 (LSTMBlockCell of the decoder)."""
    from tensorflow_end2end_speech_recognition_amd import ops
    rng = np.random.RandomState(B + K + U)
    f = lambda *s, sc=1.0: torch.tensor(rng.randn(*s) * sc, dtype=torch.float32, device=cuda)
    wide = f(B, K + 8, sc=0.5)                               # x as a row block of a wider array
    x = wide[:, :K]
    W, b = f(K, 4 * U, sc=0.06), (f(4 * U, sc=0.2) if bias else None)
    pp = f(3, U, sc=0.3) if peep else None
    cp, hp = f(B, U, sc=2.0), f(B, U, sc=0.5)
    live = torch.tensor((rng.rand(B) < 0.7).astype(np.float32), device=cuda)
    live[0] = 1.0
    om = torch.tensor((rng.rand(B, U) < 0.8) / 0.8, dtype=torch.float32, device=cuda) if mask else None
    next_in, av = torch.zeros(B, K, device=cuda), torch.zeros(B, U + 24, device=cuda)
    next_in2, av2 = torch.zeros_like(next_in), torch.zeros_like(av)
    pre = ops.gemm(x, W, bias=b)
    want = ops.lstm_cell_fwd(pre, cp, hp, pp, live, 1.0, 3.0, out_mask=om, want_cell_out=True,
                             h_also=next_in[:, K - U:], cell_out_also=av[:, :U])
    W_il = ops.lstm_cell_gemm_prep(W, b)
    assert torch.equal(W_il[:K].view(K, U, 4), W.view(K, 4, U).transpose(1, 2))
    got = ops.lstm_cell_gemm_fwd(x, W_il, bias, cp, hp, pp, live, 1.0, 3.0, out_mask=om, h_also=next_in2[:, K - U:],
                                 cell_out_also=av2[:, :U])
    for name, g, w in zip(('gates', 'c_raw', 'c_out', 'h_out', 'h_raw', 'cell_out'), got, want):
        assert torch.equal(g, w), (name, float((g - w).abs().max()))
    assert torch.equal(next_in2, next_in) and torch.equal(av2, av)
    assert float(want[0].abs().sum()) > 0 and float((want[2] - cp).abs().max()) > 0     # the clip (3.0) and live rows are exercised
    assert float(want[1].abs().max()) <= 3.0


@pytest.mark.parametrize('B,K,U', [(32, 1600, 512), (5, 192, 64), (17, 640, 128), (1, 64, 16)])
def test_cell_products_with_bf16_weight_images(cuda, B, K, U):
    """asr_lstm_cell_gemm_fwd_h / _bwd_h (bf16 weight images: fragment-ordered, gate-interleaved for the forward product +
    cell; plain rows for dpre W^T) against the fp32 kernels on the same weights rounded to bf16 -- the activations stay
    fp32 and the products are exact, so only the summation order differs: 1e-5 of the largest entry."""
    from tensorflow_end2end_speech_recognition_amd import ops
    rng = np.random.RandomState(B + K + U + 1)
    f = lambda *s, sc=1.0: torch.tensor(rng.randn(*s) * sc, dtype=torch.float32, device=cuda)
    x, W, b, pp = f(B, K, sc=0.5), f(K, 4 * U, sc=0.06), f(4 * U, sc=0.2), f(3, U, sc=0.3)
    cp, hp = f(B, U, sc=2.0), f(B, U, sc=0.5)
    live = torch.ones(B, device=cuda)
    live[B // 2:] = torch.tensor((rng.rand(B - B // 2) < 0.6).astype(np.float32), device=cuda)
    om = torch.tensor((rng.rand(B, U) < 0.8) / 0.8, dtype=torch.float32, device=cuda)
    Wr = W.to(torch.bfloat16).float()
    img = ops.lstm_cell_gemm_prep_h(W, b)
    nxt, av = torch.zeros(B, K, device=cuda), torch.zeros(B, U + 8, device=cuda)
    nxt2, av2 = torch.zeros_like(nxt), torch.zeros_like(av)
    want = ops.lstm_cell_fwd(ops.gemm(x, Wr, bias=b), cp, hp, pp, live, 1.0, 3.0, out_mask=om, want_cell_out=True,
                             h_also=nxt[:, K - U:], cell_out_also=av[:, :U])
    got = ops.lstm_cell_gemm_fwd_h(x, img, U, cp, hp, pp, live, 1.0, 3.0, out_mask=om, h_also=nxt2[:, K - U:],
                                   cell_out_also=av2[:, :U])
    rel = lambda g, w: float((g - w).abs().max() / w.abs().max().clamp_min(1e-6))
    for name, g, w in zip(('gates', 'c_raw', 'c_out', 'h_out', 'h_raw', 'cell_out'), got, want):
        assert rel(g, w) < 1e-5, (name, rel(g, w))
    assert rel(nxt2, nxt) < 1e-5 and rel(av2, av) < 1e-5
    dpre = f(B, 4 * U, sc=0.3)
    assert rel(ops.lstm_cell_gemm_bwd_h(dpre, img, K), ops.gemm(dpre, Wr, transB=True)) < 1e-5


@pytest.mark.parametrize('att,prev,sig,dtype', [('location', 'carry', False, 'f32'), ('hybrid', 'zeros', False, 'bf16'),
                                               ('bahdanau_content', 'zeros', True, 'f32'), ('luong_dot', 'zeros', False, 'f32')])
def test_native_greedy_inference_loop(cuda, att, prev, sig, dtype):
    """asr_att_decoder_infer (every decoder step, the output head, argmax, the embedding of the chosen id and the
    per-row finished flags issued from one call; one read-back at the end) against the class-surface path
    (AttentionDecoder.step under dynamic_decode, a device sync per token) and against oracle.attention: ids identical,
    same number of emitted steps.  The end-of-sequence bias is set so that (a) no row ever finishes (all 40 steps), (b) rows
    finish at different steps (imputed zeros behind the first EOS, state copy-through, zeroed context input), (c) every
    row finishes at once (one step; the early exit must not change the result).  Reference:
    models/attention/attention_seq2seq.py:462-509, decoders/dynamic_decoder.py:148-197."""
    from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
    rng = np.random.RandomState(5)
    B, T, D, H, L, U, A, Em, C = 5, 70, 12, 64, 1, 128, 32, 8, 9
    if att == 'luong_dot':
        U = 2 * H
    x, sl, labels, lsl, _ = _batch(rng, B, T, D, C)
    model = AttentionSeq2Seq(input_size=D, encoder_type='blstm', encoder_num_units=H, encoder_num_layers=L,
                             encoder_num_proj=None, attention_type=att, attention_dim=A, decoder_type='lstm',
                             decoder_num_units=U, decoder_num_layers=1, embedding_dim=Em, num_classes=C, sos_index=C,
                             eos_index=C + 1, max_decode_length=40, parameter_init=0.1, clip_grad_norm=5.0,
                             clip_activation_encoder=50, clip_activation_decoder=50, dtype=dtype, seed=5,
                             sharpening_factor=1.5, sigmoid_smoothing=sig, prev_alpha=prev)
    lens = []
    for eos_bias in (-50.0, 0.35, 50.0):
        sd = {k: v.clone() for k, v in model.store.state_dict().items()}
        sd['attention_decoder/decoder/output_layer/biases'][C + 1] = eos_bias
        model.store.load_state_dict(sd)
        ids_native = model.infer(x, sl, native=True)
        raw = model._infer_raw
        ids_class = model.infer(x, sl, native=False)
        assert ids_native.shape == ids_class.shape and np.array_equal(ids_native, ids_class), (eos_bias, ids_native, ids_class)
        assert raw['steps'] == ids_native.shape[1] and raw['steps_issued'] >= raw['steps']
        lens.append(ids_native.shape[1])
        if dtype == 'f32':
            sdn = {k: v.cpu().numpy() for k, v in sd.items()}
            ref_ids = oatt.attention_model_infer(sdn, x, sl, L, att, C, C + 1, 40, clip_enc=50.0, clip_dec=50.0,
                                                 sharpening=1.5, sigmoid_smoothing=sig, prev_alpha=prev)
            assert np.array_equal(ids_native, ref_ids), eos_bias
    assert lens[0] == 40 and lens[2] == 1 and 1 <= lens[1] <= 40


@pytest.mark.parametrize('case', ['location_zeros_bf16', 'bahdanau_sigmoid_f32', 'location_carry_bf16', 'hybrid_carry_f32',
                                  'luong_dot_small', 'bahdanau_sigmoid_f32/one_step', 'location_carry_bf16/one_step',
                                  'location_zeros_bf16/em64', 'hybrid_carry_f32/em64'])
def test_native_decoder_loop_against_the_step_by_step_statement(cuda, monkeypatch, case):
    """asr_att_decoder_fwd / _bwd (all To steps from one call, with the fused kernels the loop uses: dctx add inside the
    4-frame d-alpha kernel, softmax backward folded into the energy backward through partial alpha.dalpha sums, dropout
    mask and carried-dh add inside the cell backward, length-limited vectorised energy / location kernels with exp2-rcp
    tanh, skinny MFMA products) against tests/_cpu_ops.py's step-by-step float64 statement of the same loop -- at
    widths that select those kernels (A = 128 -> 32 lanes x float4 per frame, 2H = 512 bf16 / 256 fp32 -> 16-byte
    encoder vectors, T = 200 -> four 64-frame chunks), which the small model-level parity tests do not reach.  The last
    case (A = U = 2H = 64, dot-product scoring without a query FC) takes the general fallbacks instead."""
    import _cpu_ops as cpu
    from tensorflow_end2end_speech_recognition_amd import ops
    one_step = case.endswith('/one_step')                    # a single decoder step: no "next step" operands anywhere
    em64 = case.endswith('/em64')      # cell input width % 64 == 0: product + cell as ONE launch (asr_lstm_cell_gemm_fwd)
    case = case.split('/')[0]
    cfg = dict(location_zeros_bf16=dict(keys=False, carry=0, sig=False, bf16=True, A=128, E2=512, mode=0, hasq=1, taps=0),
               bahdanau_sigmoid_f32=dict(keys=True, carry=0, sig=True, bf16=False, A=128, E2=256, mode=0, hasq=1, taps=0),
               location_carry_bf16=dict(keys=False, carry=1, sig=False, bf16=True, A=128, E2=512, mode=0, hasq=1, taps=201),
               hybrid_carry_f32=dict(keys=True, carry=1, sig=True, bf16=False, A=32, E2=256, mode=0, hasq=1, taps=200),
               luong_dot_small=dict(keys=True, carry=0, sig=False, bf16=False, A=64, E2=64, mode=1, hasq=0, taps=0))[case]
    rng = np.random.RandomState(len(case))
    B, T, To, Em = 3, 200, (1 if one_step else 5), (64 if em64 else 8)
    A, E2 = cfg['A'], cfg['E2']
    U = A if not cfg['hasq'] else 64
    Din = Em + E2 + U
    f = lambda *s, sc=1.0: torch.tensor(rng.randn(*s) * sc, dtype=torch.float32)
    seq_len = torch.tensor([T, 77, 130], dtype=torch.int32)
    live = torch.tensor([[1, 1, 1], [1, 1, 1], [1, 0, 1], [1, 0, 1], [1, 0, 0]], dtype=torch.float32)[:To].contiguous()   # [To,B]
    enc = f(T, B, E2, sc=0.5)
    enc = enc * (torch.arange(T).view(T, 1, 1) < seq_len.view(1, B, 1))
    if cfg['bf16']:
        enc = enc.to(torch.bfloat16)
    a = dict(To=To, B=B, T=T, U=U, Em=Em, E2=E2, A=A, att_mode=cfg['mode'], has_query_fc=cfg['hasq'],
             carry_alpha=cfg['carry'], taps=cfg['taps'], enc_dtype=1 if cfg['bf16'] else 0, forget_bias=1.0,
             cell_clip=3.0, sharpening=1.5,
             W_cell=f(Din, 4 * U, sc=0.08), b_cell=f(4 * U, sc=0.1), peep=f(3, U, sc=0.1),
             W_q=f(U, A, sc=0.1) if cfg['hasq'] else None, b_q=f(A, sc=0.1) if (cfg['hasq'] and cfg['carry']) else None,
             v=f(A, sc=0.5) if cfg['mode'] == 0 else None,
             keys=(enc.float() if case == 'luong_dot_small' else f(T, B, A, sc=0.5)) if cfg['keys'] else None,
             enc=enc, seq_len=seq_len,
             filt=f(cfg['taps'], 1, 10, sc=0.3) if cfg['carry'] else None, wfil=f(10, A, sc=0.3) if cfg['carry'] else None,
             alpha_zero=torch.zeros(B, T) if cfg['carry'] else None, live=live,
             dmask=(torch.tensor(rng.rand(To, B, U) < 0.8, dtype=torch.float32) / 0.8),
             dec_in=f(To, B, Din, sc=0.5), av_in=torch.zeros(To, B, U + E2), alpha_all=torch.zeros(To, B, T),
             snorm_all=torch.zeros(To, B) if cfg['sig'] else None, gates_all=torch.zeros(To, B, 4 * U),
             craw_all=torch.zeros(To, B, U), c_all=torch.zeros(To + 1, B, U), h_all=torch.zeros(To + 1, B, U),
             qz_all=torch.zeros(To, B, A))
    a['c_all'][0] = f(B, U, sc=0.3)
    a['h_all'][0] = f(B, U, sc=0.3)
    a['dec_in'][0, :, Em:Em + E2] = 0.0                      # no context before the first step
    a['dec_in'][0, :, Em + E2:] = a['h_all'][0]
    bwd_in = dict(dav_cell=f(To, B, U, sc=0.3), dav_ctx=f(To, B, E2, sc=0.3))
    bwd_out = dict(dctx_all=(To, B, E2), dpre_all=(To, B, 4 * U), dqz_all=(To, B, A),
                   dv_all=(To, B, A) if cfg['mode'] == 0 else None, dpeep_all=(To, B, 3 * U), d_in_all=(To, B, Din),
                   dkeys=(T, B, A) if cfg['keys'] else None, dwfil_rows=(B, 10, A) if cfg['carry'] else None,
                   dfilt_rows=(B, cfg['taps'], 10) if cfg['carry'] else None, dc0=(B, U), dh0=(B, U))

    def clone_to(dev):
        d = {k: (v.clone().to(dev) if torch.is_tensor(v) else v) for k, v in a.items()}
        d.update({k: v.clone().to(dev) for k, v in bwd_in.items()})
        d.update({k: (torch.zeros(s, device=dev) if s is not None else None) for k, s in bwd_out.items()})
        return d

    ref, got = clone_to('cpu'), clone_to(cuda)
    cpu._att_decoder_fwd(ref)
    ops.att_decoder_fwd(got)
    assert (got.get('W_cell_il') is not None) == em64
    if em64:       # the one-launch product + cell against the two launches it replaces: the whole loop bit for bit
        two = clone_to(cuda)
        monkeypatch.setattr(ops, 'FUSED_CELL_GEMM', False)
        ops.att_decoder_fwd(two)
        assert two.get('W_cell_il') is None
        for name in ('alpha_all', 'av_in', 'dec_in', 'c_all', 'h_all', 'qz_all', 'gates_all', 'craw_all'):
            assert torch.equal(got[name], two[name]), name
    # relative to the array's largest entry, with a floor: without keys and without carried location features every frame
    # has the same energy, so dqz / dv are (sum_t denergy) x const = rounding noise around zero (1e-7) on both sides
    rel = lambda x, y: float(np.abs(x.cpu().double().numpy() - y.double().numpy()).max() / max(np.abs(y.double().numpy()).max(), 1e-2))
    for name in ('alpha_all', 'av_in', 'dec_in', 'c_all', 'h_all', 'qz_all', 'gates_all', 'craw_all') + (('snorm_all',) if cfg['sig'] else ()):
        assert rel(got[name], ref[name]) < 2e-5, (name, rel(got[name], ref[name]))
    # the backward of both sides starts from the SAME saved forward (the reference's), so the comparison is per kernel
    for name in ('alpha_all', 'av_in', 'dec_in', 'c_all', 'h_all', 'qz_all', 'gates_all', 'craw_all', 'snorm_all'):
        if ref.get(name) is not None:
            got[name].copy_(ref[name].to(cuda))
    cpu._att_decoder_bwd(ref)
    ops.att_decoder_bwd(got)
    for name, shape in bwd_out.items():
        if shape is not None:
            assert rel(got[name], ref[name]) < 5e-5, (name, rel(got[name], ref[name]))
    assert ops.check_async_errors(0) == 0


@pytest.mark.parametrize('att,prev,sig', [('bahdanau_content', 'zeros', False), ('location', 'carry', False),
                                          ('hybrid', 'carry', True), ('luong_dot', 'zeros', False),
                                          ('luong_concat', 'zeros', False)])
def test_attention_class_surface(cuda, att, prev, sig):
    """AttentionLayer / LSTMDecoderCell / AttentionDecoder under dynamic_decode with a TrainingHelper, and
    InitialStateBridge (models/attention/decoders/*.py, bridge.py) on the HIP kernels: teacher-forced logits / weights /
    ids equal the oracle's and the model's own fused loop (asr_att_decoder_fwd); a standalone AttentionLayer equals
    oracle.attention.attention_step.  (Greedy inference -- the path these classes serve in the model -- is checked
    bit-exact against the oracle by every test_attention_model_parity case.)"""
    import _config_parity as cp
    r = cp.run_class_surface('cuda:0', att, prev, sig, B=5, T=70, To=6, D=12, H=64, L=1, U=128, A=32, Em=8, C=9)
    assert r['loss_rel'] < 1e-4, r['report']
    assert r['class_logits_vs_oracle'] < 2e-4 and r['class_alpha_vs_oracle'] < 1e-5 and r['class_ids_vs_oracle'] == 0, r['report']
    assert r['class_logits_vs_fused'] < 2e-4 and r['class_alpha_vs_fused'] < 1e-5, r['report']
    assert r['layer_alpha'] < 1e-5 and r['layer_ctx'] < 2e-4, r['report']


def test_bridge_classes(cuda):
    import _config_parity as cp
    r = cp.run_bridges('cuda:0')
    assert r['fc_err'] < 1e-5 and r['ok_zero'] and r['ok_pass'] and r['raised'], r
