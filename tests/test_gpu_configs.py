"""GPU parity of BASELINE configs[2], [3], [4] AS MODELS, at their own widths (SURVEY 8d cfg C / D / E), against the
oracle -- the HIP path through the C ABI on one side, oracle/model.py / oracle/attention.py on the other, same
parameters, same seeded ragged batch.  bf16-operand models are compared with the oracle evaluated at the device path's
rounding points (`operand_round`: inputs, MFMA weight operands, every stored activation / emitted h; straight-through,
state and accumulation fp64), so the bounds state arithmetic, not the precision choice; the fp32 run of the decoder
widths has the plain fp64 oracle and the 1e-4 bar of north_star.

The runs themselves live in tests/_config_parity.py (also executed on CPU stand-ins at toy widths by the CPU suite).
Sequence lengths are cut to what the fp64 oracle finishes in about a minute; the widths -- which select the kernels --
are the configurations' own.
"""
import numpy as np
import pytest

import _config_parity as cp

pytestmark = pytest.mark.gpu


def _no_handoff_errors():
    from tensorflow_end2end_speech_recognition_amd import ops
    assert ops.check_async_errors(0) == 0


def test_cfgC_vgg_blstm_4x512_bf16_ragged_two_tiles(cuda):
    """configs[2]: VGG (40 mel x splice 11 x 3 images, 64 / 128-channel implicit-GEMM convolutions on MFMA) -> bridge ->
    4 x 512 BLSTM (8-CU cluster kernels) -> 29-class CTC, bf16 operands, B = 20 ragged utterances: two 16-utterance
    recurrence tiles (12 rows of the second are padding) and the valid-frame gather in front of the convolutions.
    Reference: models/encoders/core/vgg_blstm.py:77-220, cnn_util.py:13-84, models/ctc/ctc.py:175-323."""
    r = cp.run_cfgC('cuda:0', 'bf16', B=20, T=60, F=40, W=11, H=512, L=4, C=28, perturb_eps=2e-6)
    print('\n' + r['report'])
    _no_handoff_errors()
    # measured on MI355X, two realisations of the same arithmetic: with the first convolution on the vector ALUs loss 5e-5,
    # per-utterance 1.7e-3, logits 3.0e-2 abs (|logit| <= 3.6), gradients of the matrices / biases 0.4e-2 .. 4.7e-2 of
    # their largest entry (relative L2 <= 3.1e-2), peephole vectors up to 9.6e-2 (6.9e-2); with it on the matrix cores --
    # 0.001 % of its bf16 outputs differ by one ulp, both forms equally close to the fp64 convolution
    # (scripts/probe_smallc.py) -- loss 1.8e-4, matrices up to 7.5e-2 (4.1e-2), peepholes 7.6e-2 (5.8e-2).  The worst entry
    # is always layer 1, which sits under four layers of bf16 BPTT and the VGG stack's bf16 activations: a rounding flip
    # there is amplified, so the bounds state that amplification, not a kernel's arithmetic.  The exact two-tile /
    # valid-frame logic is what the fp32 run below pins to 2e-3.
    assert r['loss_rel'] < 2e-3 and r['per_utt_rel'] < 5e-3, r['report']
    assert r['logits_abs'] < 3e-2 * max(1.0, r['logits_max']), r['report']
    # Round 4: the amplification is SHOWN, not asserted.  The same oracle with every rounding point nudged by a relative
    # 2e-6 (a fraction ~5e-4 of the roundings flip by one bf16 ulp -- what two correct summation orders differ by) moves
    # its own worst gradient entry by 12.5e-2 (matrices, L2 10.9e-2) / 12.8e-2 (peepholes) and the loss by 1.3e-3: more than
    # the device-vs-oracle gap (7.5e-2 / 4.2e-2 / 7.6e-2, loss 1.8e-4).  With fp32 operands the same model at the same
    # widths agrees to 2e-3 (test_cfgC_4x512_fp32_operands_two_tiles).  Bounds: 1.5 x measured, and never beyond what the
    # nudged oracle shows rounding alone does.
    assert r['grad_worst_matrices'] < 1.13e-1 and r['grad_worst_l2'] < 6.3e-2, r['report']
    assert r['grad_worst_peepholes'] < 1.14e-1 and r['grad_worst_l2_peepholes'] < 8.7e-2, r['report']
    assert r['grad_worst_matrices'] < r['perturb_worst_matrices'] and r['grad_worst_l2'] < r['perturb_worst_l2'], r['report']
    assert r['grad_worst_peepholes'] < r['perturb_worst_peepholes'] and r['loss_rel'] < r['perturb_loss_rel'], r['report']


def test_cfgC_two_tiles_ragged_fp32(cuda):
    """The same configs[2] topology with fp32 operands end to end (exact-fp32 MFMA products, im2col convolutions):
    B = 20 ragged -> two recurrence tiles with their per-tile bias / peephole partial sums reduced, valid-frame gather
    in front of the VGG stack -- against the plain fp64 oracle at the fp32 bars (loss 1e-4, gradients 2e-3)."""
    r = cp.run_cfgC('cuda:0', 'f32', B=20, T=40, F=40, W=11, H=256, L=2, C=28)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['per_utt_rel'] < 1e-4, r['report']
    assert r['logits_abs'] < 2e-4 * max(1.0, r['logits_max']), r['report']
    assert r['grad_worst'] < 2e-3, r['report']


def test_cfgC_4x512_fp32_operands_two_tiles(cuda):
    """configs[2] at its OWN widths (VGG on 40 x 11 x 3 images, 4 x 512 BLSTM on the fp32 cluster kernels, two ragged
    recurrence tiles) with fp32 operands end to end against the plain fp64 oracle at the fp32 bars: with no bf16
    rounding anywhere the whole model -- the same host wiring, gather, tile reduction and layer stack the bf16 test
    runs -- agrees to 2e-3 of every gradient's largest entry.  What the bf16 run adds on top is rounding."""
    r = cp.run_cfgC('cuda:0', 'f32', B=20, T=60, F=40, W=11, H=512, L=4, C=28)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['per_utt_rel'] < 1e-4, r['report']
    assert r['logits_abs'] < 2e-4 * max(1.0, r['logits_max']), r['report']
    assert r['grad_worst'] < 2e-3, r['report']


def test_cfgC_long_sequences_bf16(cuda):
    """configs[2] at its widths on LONG utterances: T = 400 frames (the bench batch has 150 .. 1650; the fp64 oracle
    takes about a minute here), B = 17 -> two recurrence tiles, ragged.  Forward quantities: loss, per-utterance losses,
    logits.  Gradients: the whole gradient as one vector and every matrix / bias in relative L2 -- per-entry figures of a
    400-step bf16 BPTT under four layers are dominated by single rounding flips (the nudged-oracle figures in
    test_cfgC_vgg_blstm_4x512_bf16_ragged_two_tiles: 12 % at T = 60 from flipping 5e-4 of the roundings), and the
    peephole vectors -- sums over 6 000 frames of products with cell states up to the clip, |g| an order below the
    matrices' -- lose all significance per variable here; the recurrence kernels' own peephole gradients are held to 6e-3
    at T = 778 by test_lstm_cluster_bf16_gradient_parity_headline_shapes.  Measured: loss 1.3e-4, per-utterance 8e-4,
    logits 9e-2 abs of |logit| <= 4.2, whole gradient 1.2e-1, worst matrix 1.7e-1 (VGG1/conv1, under everything)."""
    r = cp.run_cfgC('cuda:0', 'bf16', B=17, T=400, F=40, W=11, H=512, L=4, C=28, seed=22)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 4e-4 and r['per_utt_rel'] < 2e-3, r['report']                  # bounds: 1.5 - 3 x measured
    assert r['logits_abs'] < 3.3e-2 * max(1.0, r['logits_max']), r['report']
    assert r['grad_global_l2'] < 1.7e-1 and r['grad_worst_l2'] < 2.5e-1, r['report']


def test_cfgD_long_sequences_bf16(cuda):
    """configs[3] at its widths on LONG utterances: T = 800 encoder frames (thirteen 64-frame chunks of the scoring
    kernels), 150 decoder steps through asr_att_decoder_fwd / _bwd with carried location features, B = 6 ragged.
    Measured: loss 1.4e-5, sequence loss 1.5e-4, CTC per-utterance 3.4e-4, attention weights 6e-9, logits 2.3e-2 abs of
    |logit| <= 3.8, 4 of 900 teacher-forced argmax ids differ (bf16 ties), whole gradient 1.5e-2, worst variable 3.2e-2
    in relative L2 (the attention layer's v_a)."""
    r = cp.run_attention('cuda:0', 'bf16', 'location', B=6, T=800, To=150, D=240, H=512, L=5, U=512, A=128, Em=64, C=28,
                         lam=0.5, prev_alpha='carry', seed=34)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['seq_loss_rel'] < 5e-4 and r['ctc_losses_rel'] < 1e-3, r['report']
    assert r['alpha_abs'] < 1e-6 and r['logits_abs'] < 1.2e-2 * max(1.0, r['logits_max']), r['report']
    assert r['grad_global_l2'] < 2.3e-2 and r['grad_worst_l2'] < 5e-2, r['report']


@pytest.mark.parametrize('prev_alpha', ['zeros', 'carry'])
def test_cfgD_joint_location_5x512_bf16(cuda, prev_alpha):
    """configs[3]: 5 x 512 BLSTM encoder (bf16 operands) on D = 240 (120 x stack 2) -> bridge -> LSTM decoder U = 512
    with LOCATION attention A = 128, embedding 64, 30 output classes, + the 29-class CTC head on the encoder's bf16
    operand copy, lambda = 0.5; B = 6, T = 200 (four 64-frame chunks of the scoring kernels), 40 decoder steps through
    asr_att_decoder_fwd / _bwd.  prev_alpha='zeros' is the reference's effective graph (quirk Q1), 'carry' the
    recurrence attention_layer.py:231-265 expresses (conv1d over the previous weights -> W_filter inside the energy
    kernels).  Loss, sequence loss, logits, attention weights, CTC logits / per-utterance losses, EVERY gradient.
    Reference: models/attention/joint_ctc_attention.py:237-346, attention_layer.py:191-265, attention_decoder.py:142-295."""
    r = cp.run_attention('cuda:0', 'bf16', 'location', B=6, T=200, To=40, D=240, H=512, L=5, U=512, A=128, Em=64, C=28,
                         lam=0.5, prev_alpha=prev_alpha)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 2e-3 and r['seq_loss_rel'] < 2e-3 and r['ctc_losses_rel'] < 5e-3, r['report']
    assert r['alpha_abs'] < 2e-3, r['report']
    assert r['logits_abs'] < 3e-2 * max(1.0, r['logits_max']), r['report']
    assert r['ctc_logits_abs'] < 3e-2 * max(1.0, r['logits_max']), r['report']
    # measured: loss 7e-5, logits 2.8e-2 abs, alpha 2e-10 (zeros) / 1.2e-7 (carry), matrices <= 2e-2, peephole vectors
    # <= 3.5e-2 of their largest entry
    assert r['grad_worst_matrices'] < 4e-2 and r['grad_worst_peepholes'] < 8e-2 and r['grad_worst_l2'] < 4e-2, r['report']


def test_cfgD_decoder_widths_fp32_directly_against_the_oracle(cuda):
    """The decoder loop at the widths of configs[3] (A = 128 -> 32 lanes x float4 per frame, U = 512 -> the skinny MFMA
    products, T = 200 -> multi-chunk scoring, carried location features) with fp32 operands end to end -- so nothing
    but kernel arithmetic separates the HIP path from oracle/attention.py: the 1e-4 loss bar of north_star, attention
    weights to 1e-5, teacher-forced ids identical, every gradient to 2e-3 of its maximum.  (The encoder is 2 x 128:
    fp32 at H = 512 would only add minutes of single-CU recurrence to a test about the decoder.)"""
    r = cp.run_attention('cuda:0', 'f32', 'location', B=6, T=200, To=40, D=240, H=128, L=2, U=512, A=128, Em=64, C=28,
                         lam=0.5, prev_alpha='carry')
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['seq_loss_rel'] < 1e-4 and r['ctc_losses_rel'] < 1e-4, r['report']
    assert r['alpha_abs'] < 1e-5 and r['ids_mismatch'] == 0, r['report']
    assert r['grad_worst'] < 2e-3, r['report']


def test_cfgD_5x512_fp32_at_its_own_widths_against_the_plain_oracle(cuda):
    """configs[3] at its OWN widths -- 5 x 512 BLSTM encoder (fp32 cluster recurrences: exact-fp32 MFMA), LOCATION
    attention A = 128, decoder U = 512, embedding 64, 30 classes + the 29-class CTC head, lambda 0.5, carried previous
    weights -- with fp32 operands end to end against the PLAIN fp64 oracle (no rounding points): the parity statement of
    north_star (loss within 1e-4 relative in fp32) at the configuration's widths rather than on a narrower encoder
    (VERDICT r05 missing 3); T is what is cut (160 frames, three scoring chunks; 30 decoder steps), not the widths.
    Bars: joint loss / sequence loss / per-utterance CTC <= 1e-4, attention weights <= 1e-5, teacher-forced ids
    identical, every gradient <= 2e-3 of its largest entry.
    Reference: models/attention/joint_ctc_attention.py:237-346, attention_layer.py:191-265, attention_decoder.py:142-295."""
    r = cp.run_attention('cuda:0', 'f32', 'location', B=6, T=160, To=30, D=240, H=512, L=5, U=512, A=128, Em=64, C=28,
                         lam=0.5, prev_alpha='carry', seed=36)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['seq_loss_rel'] < 1e-4 and r['ctc_losses_rel'] < 1e-4, r['report']
    assert r['alpha_abs'] < 1e-5 and r['ids_mismatch'] == 0, r['report']
    assert r['grad_worst'] < 2e-3, r['report']


def test_cfgE_hybrid_kanji_5x512_fp32_at_its_own_widths_against_the_plain_oracle(cuda):
    """configs[4] at its OWN widths with fp32 operands end to end: 5 x 512 encoder on D = 246, HYBRID attention, the
    3 388-class attention softmax and the 3 387-class CTC head, against the plain fp64 oracle at the fp32 bars (the
    bf16 run of this configuration is held to the oracle at bf16 rounding points; this is the run that carries the
    1e-4 statement).  B = 4, T = 120, 20 decoder steps.
    Reference: models/attention/attention_layer.py:191-229, joint_ctc_attention.py:182-346."""
    r = cp.run_attention('cuda:0', 'f32', 'hybrid', B=4, T=120, To=20, D=246, H=512, L=5, U=512, A=128, Em=64, C=3386,
                         lam=0.5, prev_alpha='zeros', seed=37)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['seq_loss_rel'] < 1e-4 and r['ctc_losses_rel'] < 1e-4, r['report']
    assert r['alpha_abs'] < 1e-5 and r['ids_mismatch'] == 0, r['report']
    assert r['grad_worst'] < 2e-3, r['report']


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_cfgD_two_pipelines_joint_location_two_tiles(cuda, dtype):
    """configs[3] with TWO 16-utterance tiles (B = 20 -> 32 padded rows): the encoder runs its two half-batch pipelines
    (models/encoders/core/blstm.py ENC_HALVES; measured slower than one pipeline and off by default, kept correct) -- two
    recurrence launches per layer side by side on two streams / two
    handles, final states of both parts into the bridge, the decoder's and the CTC head's gradients back into both
    parts, weight gradients accumulated across them -- under the joint location model at the configuration's widths.
    fp32: the plain oracle at the fp32 bars; bf16: the oracle at the device path's rounding points, bounds of
    test_cfgD_joint_location_5x512_bf16."""
    r = cp.run_attention('cuda:0', dtype, 'location', B=20, T=120, To=24, D=240, H=512, L=5, U=512, A=128, Em=64, C=28,
                         lam=0.5, prev_alpha='carry', seed=38, halves=True)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['model'].encoder._split == 16 and r['model'].encoder.layers_b is not None
    if dtype == 'f32':
        assert r['loss_rel'] < 1e-4 and r['seq_loss_rel'] < 1e-4 and r['ctc_losses_rel'] < 1e-4, r['report']
        assert r['alpha_abs'] < 1e-5 and r['ids_mismatch'] == 0, r['report']
        assert r['grad_worst'] < 2e-3, r['report']
    else:
        assert r['loss_rel'] < 2e-3 and r['seq_loss_rel'] < 2e-3 and r['ctc_losses_rel'] < 5e-3, r['report']
        assert r['alpha_abs'] < 2e-3 and r['logits_abs'] < 3e-2 * max(1.0, r['logits_max']), r['report']
        # (measured at B = 20: worst matrix entry 4.0e-2 of its maximum, relative L2 <= 2.4e-2 -- 2.0e-2 / 3.5e-2 at B = 6)
        assert r['grad_worst_matrices'] < 6e-2 and r['grad_worst_peepholes'] < 8e-2 and r['grad_worst_l2'] < 4e-2, r['report']


def test_cfgC_two_pipelines_vgg_two_tiles_fp32(cuda):
    """The two half-batch pipelines under the VGG front-end (the input gradient of both parts back into the convolutions),
    fp32 operands against the plain oracle at the fp32 bars."""
    r = cp.run_cfgC('cuda:0', 'f32', B=20, T=40, F=40, W=11, H=256, L=2, C=28, halves=True)
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 1e-4 and r['per_utt_rel'] < 1e-4 and r['grad_worst'] < 2e-3, r['report']


def test_cfgE_hybrid_kanji_vocabulary_bf16(cuda):
    """configs[4]: 5 x 512 encoder on D = 246 (123 x stack 2), HYBRID attention (projected keys + location term),
    a 3 388-class attention softmax (3 386 kanji + SOS + EOS) and the 3 387-class CTC head whose three [T*B x 2H x C]
    products run on bf16 operands; B = 4, T = 120, 20 decoder steps.
    Reference: models/attention/attention_layer.py:191-229, joint_ctc_attention.py:182-346."""
    r = cp.run_attention('cuda:0', 'bf16', 'hybrid', B=4, T=120, To=20, D=246, H=512, L=5, U=512, A=128, Em=64, C=3386,
                         lam=0.5, prev_alpha='zeros')
    print('\n' + r['report'])
    _no_handoff_errors()
    assert r['loss_rel'] < 2e-3 and r['seq_loss_rel'] < 2e-3 and r['ctc_losses_rel'] < 5e-3, r['report']
    assert r['alpha_abs'] < 2e-3, r['report']
    assert r['logits_abs'] < 3e-2 * max(1.0, r['logits_max']), r['report']
    # measured: loss 6e-5, logits 3.9e-2 abs (|logit| <= 5.1), alpha 1.8e-4, gradients <= 1.6e-2
    assert r['grad_worst_matrices'] < 4e-2 and r['grad_worst_peepholes'] < 8e-2 and r['grad_worst_l2'] < 4e-2, r['report']
