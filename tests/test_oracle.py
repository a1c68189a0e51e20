"""CPU: pin the oracle against the reference's own golden vectors and independent
second implementations (torch.nn.LSTM, torch ctc_loss, finite differences)."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import decoders as odec
from oracle import lstm as olstm
from oracle import model as omodel
from oracle import optim as oopt

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_decoders_match_reference_golden():
    g = np.load(os.path.join(GOLD, 'decoders_v1.npz'))
    n = int(g['num_cases'])
    assert n >= 20
    for i in range(n):
        probs, sl = g['c%d_probs' % i], g['c%d_seq_len' % i]
        C = probs.shape[2]
        lp = np.log(probs)
        assert odec.greedy_decode(lp, sl, C - 1)[0] == list(g['c%d_greedy' % i])
        for w in g['c%d_widths' % i]:
            hyp, score = odec.beam_search_decode(lp, sl, C - 1, int(w))
            assert hyp[0] == list(g['c%d_beam%d' % (i, w)]), (i, w)
            assert abs(score[0] - float(g['c%d_beam%d_score' % (i, w)])) < 1e-9


def _mk_lstm(D, H, seed=0):
    rng = np.random.RandomState(seed)
    return rng, olstm.init_lstm_params(rng, D, H), olstm.init_lstm_params(rng, D, H)


def test_lstm_matches_torch_nn_lstm_without_peepholes():
    B, T, D, H = 4, 11, 6, 5
    rng, p_fw, p_bw = _mk_lstm(D, H)
    for p in (p_fw, p_bw):
        for k in ('wci', 'wcf', 'wco'):
            p[k].zero_()
        p['b'] = torch.tensor(rng.uniform(-.1, .1, 4 * H))
    x = torch.tensor(rng.randn(B, T, D))
    sl = torch.tensor([11, 7, 4, 1])
    out, fin = olstm.blstm_layer(x.transpose(0, 1), sl, p_fw, p_bw)
    m = torch.nn.LSTM(D, H, bidirectional=True).double()

    def load(p, sfx):
        def reorder(mat):  # i,g,f,o -> torch i,f,g,o
            i, g, f, o = mat[:, :H], mat[:, H:2 * H], mat[:, 2 * H:3 * H], mat[:, 3 * H:]
            return torch.cat([i, f, g, o], 1)
        wr = reorder(p['w'])
        fb = torch.cat([torch.zeros(2 * H), torch.ones(H), torch.zeros(H)]).double()
        br = reorder((p['b'] + fb).unsqueeze(0))[0]
        getattr(m, 'weight_ih_l0' + sfx).data = wr[:D].t().contiguous()
        getattr(m, 'weight_hh_l0' + sfx).data = wr[D:].t().contiguous()
        getattr(m, 'bias_ih_l0' + sfx).data = br
        getattr(m, 'bias_hh_l0' + sfx).data.zero_()
    load(p_fw, '')
    load(p_bw, '_reverse')
    pk = torch.nn.utils.rnn.pack_padded_sequence(x.transpose(0, 1), sl, enforce_sorted=True)
    o, (hn, cn) = m(pk)
    o, _ = torch.nn.utils.rnn.pad_packed_sequence(o, total_length=T)
    assert (o - out).abs().max() < 1e-12
    assert (hn[0] - fin[0][1]).abs().max() < 1e-12 and (cn[1] - fin[1][0]).abs().max() < 1e-12


def test_lstm_peephole_cell_equations():
    """models/recurrent/layers/lstm.py:142-170 written out by hand for one step."""
    rng = np.random.RandomState(3)
    D, H, B = 3, 2, 2
    p = olstm.init_lstm_params(rng, D, H, init=0.5)
    x, c0, h0 = (torch.tensor(rng.randn(B, n)) for n in (D, H, H))
    c1, h1 = olstm.lstm_block_cell(x, c0, h0, p['w'], p['b'], p['wci'], p['wcf'], p['wco'], 1.0, 0.3)
    z = np.concatenate([x.numpy(), h0.numpy()], 1) @ p['w'].numpy() + p['b'].numpy()
    sig = lambda v: 1 / (1 + np.exp(-v))
    i, g, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    c = sig(f + 1.0 + p['wcf'].numpy() * c0.numpy()) * c0.numpy() + \
        sig(i + p['wci'].numpy() * c0.numpy()) * np.tanh(g)
    c = np.clip(c, -0.3, 0.3)
    h = sig(o + p['wco'].numpy() * c) * np.tanh(c)
    assert np.abs(c1.numpy() - c).max() < 1e-12 and np.abs(h1.numpy() - h).max() < 1e-12


def test_padded_frames_are_zero_and_state_carried():
    rng, p_fw, p_bw = _mk_lstm(4, 3, 1)
    x = torch.tensor(rng.randn(2, 6, 4))
    sl = torch.tensor([6, 2])
    out, fin = olstm.blstm_layer(x.transpose(0, 1), sl, p_fw, p_bw)
    assert out[2:, 1].abs().max() == 0
    # fw final state of the short utterance == its output at its last valid frame
    assert torch.allclose(fin[0][1][1], out[1, 1, :3])
    assert torch.allclose(fin[1][1][1], out[0, 1, 3:])


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_ctc_matches_torch(seed):
    rng = np.random.RandomState(seed)
    T, B, C = 25, 5, 6
    logits = rng.randn(T, B, C) * 2
    sl = np.array([25, 20, 9, 13, 3])
    labs = [[0, 1, 1, 2], [3, 3, 3], [], [4, 0, 4, 0, 1], [2, 2]]   # last one: needs 3 frames, feasible
    loss, grad = octc.ctc_loss_batch(logits, labs, sl)
    lt = torch.tensor(logits, requires_grad=True)
    tl = torch.nn.functional.ctc_loss(lt.log_softmax(2), torch.tensor(sum(labs, []), dtype=torch.long),
                                      torch.tensor(sl), torch.tensor([len(l) for l in labs]),
                                      blank=C - 1, reduction='none', zero_infinity=True)
    tl.sum().backward()
    assert np.abs(loss - tl.detach().numpy()).max() < 1e-10
    assert np.abs(grad - lt.grad.numpy()).max() < 1e-10
    assert np.abs(grad[sl[2]:, 2]).max() == 0


def test_ctc_infeasible_is_zero():
    rng = np.random.RandomState(0)
    logits = rng.randn(4, 1, 5)
    loss, grad = octc.ctc_loss_batch(logits, [[1, 1, 1]], np.array([4]))   # needs 5 frames
    assert loss[0] == 0 and np.abs(grad).max() == 0
    with pytest.raises(ValueError):
        octc.ctc_loss_batch(logits, [[1, 1, 1]], np.array([4]), ignore_longer=False)


def test_ctc_model_gradient_finite_difference():
    rng = np.random.RandomState(0)
    B, T, D, H, C, L = 3, 8, 6, 4, 5, 2
    sd = {}
    for i in range(1, L + 1):
        din = D if i == 1 else 2 * H
        for d in ('fw', 'bw'):
            p = olstm.init_lstm_params(rng, din, H, init=0.3)
            base = 'blstm_hidden%d/%s/lstm_cell' % (i, d)
            sd[base + '/kernel'] = p['w'].numpy()
            sd[base + '/bias'] = p['b'].numpy()
            sd[base + '/w_i_diag'], sd[base + '/w_f_diag'], sd[base + '/w_o_diag'] = \
                p['wci'].numpy(), p['wcf'].numpy(), p['wco'].numpy()
    sd['output/weights'] = rng.randn(2 * H, C) * .3
    sd['output/biases'] = np.zeros(C)
    x = rng.randn(B, T, D)
    sl = np.array([8, 5, 7])
    labs = [[0, 1], [2], [3, 3, 1]]
    r = omodel.ctc_model_forward(sd, x, labs, sl, L, weight_decay=1e-3)
    for k, idx in [('blstm_hidden1/bw/lstm_cell/kernel', (1, 2)), ('blstm_hidden2/fw/lstm_cell/w_o_diag', (1,)),
                   ('blstm_hidden1/fw/lstm_cell/w_f_diag', (0,)), ('output/weights', (3, 1)),
                   ('blstm_hidden2/bw/lstm_cell/bias', (5,))]:
        eps = 1e-6
        sd2 = {a: np.array(b, dtype=np.float64).copy() for a, b in sd.items()}
        sd2[k][idx] += eps
        r2 = omodel.ctc_model_forward(sd2, x, labs, sl, L, weight_decay=1e-3)
        fd = (r2['total_loss'] - r['total_loss']) / eps
        assert abs(fd - r['grads'][k][idx]) < 1e-5 * max(1, abs(fd)), k


def test_ctc_model_bottleneck_gradient_finite_difference():
    """bottleneck FC + ReLU (models/ctc/ctc.py:201-216) in the oracle: analytic vs finite differences."""
    rng = np.random.RandomState(1)
    B, T, D, H, C, BN = 2, 6, 5, 3, 4, 7
    sd = {}
    for d in ('fw', 'bw'):
        p = olstm.init_lstm_params(rng, D, H, init=0.3)
        base = 'blstm_hidden1/%s/lstm_cell' % d
        sd[base + '/kernel'], sd[base + '/bias'] = p['w'].numpy(), p['b'].numpy()
        sd[base + '/w_i_diag'], sd[base + '/w_f_diag'], sd[base + '/w_o_diag'] = \
            p['wci'].numpy(), p['wcf'].numpy(), p['wco'].numpy()
    sd['bottleneck/weights'] = rng.randn(2 * H, BN) * .5
    sd['bottleneck/biases'] = rng.randn(BN) * .1
    sd['output/weights'] = rng.randn(BN, C) * .3
    sd['output/biases'] = np.zeros(C)
    x, sl, labs = rng.randn(B, T, D), np.array([6, 4]), [[0, 1], [2]]
    r = omodel.ctc_model_forward(sd, x, labs, sl, 1, bottleneck=True)
    assert r['logits'].shape == (T, B, C)
    for k, idx in [('bottleneck/weights', (2, 3)), ('bottleneck/biases', (1,)), ('output/weights', (4, 2)),
                   ('blstm_hidden1/fw/lstm_cell/kernel', (0, 1))]:
        eps = 1e-6
        sd2 = {a: np.array(b, dtype=np.float64).copy() for a, b in sd.items()}
        sd2[k][idx] += eps
        fd = (omodel.ctc_model_forward(sd2, x, labs, sl, 1, bottleneck=True)['total_loss'] - r['total_loss']) / eps
        assert abs(fd - r['grads'][k][idx]) < 1e-5 * max(1, abs(fd)), k


def test_luong_concat_energy_is_the_fc_over_the_concatenation():
    """attention_layer.py:314-345: energy = v_a . tanh(W_concat [h_enc; h_dec]); the device path runs it as
    the key half and the query half of W_concat -- check the oracle states the reference form and that the
    two forms agree."""
    import torch
    from oracle import attention as oatt
    rng = np.random.RandomState(2)
    B, T, E2, U, A = 3, 5, 6, 4, 7
    enc = torch.tensor(rng.randn(B, T, E2))
    s = torch.tensor(rng.randn(B, U))
    p = {'W_concat/weights': torch.tensor(rng.randn(E2 + U, A)), 'v_a': torch.tensor(rng.randn(A))}
    sl = torch.tensor([5, 3, 4])
    alpha, ctx = oatt.attention_step(p, 'luong_concat', enc, None, s, sl)
    W = p['W_concat/weights']
    e = (p['v_a'] * torch.tanh(enc @ W[:E2] + (s @ W[E2:]).unsqueeze(1))).sum(2)
    mask = (torch.arange(T).unsqueeze(0) < sl.unsqueeze(1))
    ref = torch.softmax(torch.where(mask, e, torch.tensor(float(np.finfo(np.float32).min), dtype=e.dtype)), dim=1)
    assert torch.allclose(alpha, ref, atol=1e-12)
    assert torch.allclose(ctx, (ref.unsqueeze(2) * enc).sum(1), atol=1e-12)
    assert float(alpha[1, 3:].abs().max()) == 0.0          # masked frames get exactly zero weight


def test_optimizers_against_torch():
    rng = np.random.RandomState(0)
    p0 = rng.randn(50)
    gs = [rng.randn(50) for _ in range(5)]
    cases = {'sgd': lambda p: torch.optim.SGD([p], lr=0.1),
             'momentum': lambda p: torch.optim.SGD([p], lr=0.1, momentum=0.9),
             'adagrad': lambda p: torch.optim.Adagrad([p], lr=0.1, initial_accumulator_value=0.1, eps=0)}
    for name, mk in cases.items():
        p = p0.copy()
        s0, s1 = oopt.init_slots(name, p)
        tp = torch.tensor(p0.copy(), requires_grad=True)
        opt = mk(tp)
        for t, g in enumerate(gs, 1):
            p, s0, s1 = oopt.step(name, p, g, s0, s1, 0.1, t)
            tp.grad = torch.tensor(g)
            opt.step()
        assert np.abs(p - tp.detach().numpy()).max() < 1e-10, name
    # clip_by_norm
    g = rng.randn(10)
    assert abs(np.linalg.norm(oopt.clip_by_norm(g, 0.5)) - 0.5) < 1e-12
    assert np.allclose(oopt.clip_by_norm(g, 100.0), g)


def test_fast_cpu_port_matches_oracle():
    """oracle/fast_cpu.py (the cpu_baseline 'port' of bench.py) == the fp64 oracle."""
    from oracle import fast_cpu
    rng = np.random.RandomState(0)
    B, T, D, H, C, L = 4, 12, 6, 8, 5, 2
    sd = {}
    for i in range(1, L + 1):
        din = D if i == 1 else 2 * H
        for d in ('fw', 'bw'):
            p = olstm.init_lstm_params(rng, din, H)
            base = 'blstm_hidden%d/%s/lstm_cell' % (i, d)
            sd[base + '/kernel'], sd[base + '/bias'] = p['w'].numpy(), p['b'].numpy()
            sd[base + '/w_i_diag'], sd[base + '/w_f_diag'], sd[base + '/w_o_diag'] = \
                p['wci'].numpy(), p['wcf'].numpy(), p['wco'].numpy()
    sd['output/weights'] = rng.randn(2 * H, C) * .1
    sd['output/biases'] = np.zeros(C)
    x = rng.randn(B, T, D)
    sl = np.array([12, 7, 9, 3])
    labs = [[0, 1], [2], [3, 3, 1], [0]]
    for b in range(B):
        x[b, sl[b]:] = 0
    r = omodel.ctc_model_forward(sd, x, labs, sl, L, cell_clip=50.)
    m = fast_cpu.CpuBLSTMCTC(sd, L, cell_clip=50.)
    loss, _ = m.loss(x, labs, sl)
    loss.backward()
    assert abs(float(loss.detach()) - r['total_loss']) < 1e-5
    assert max(np.abs(m.params[k].grad.numpy() - r['grads'][k]).max() for k in sd) < 1e-5
    assert np.isfinite(m.train_step(x, labs, sl))
    # the rmsprop variant bench.py times: one step == oracle.optim on the clipped oracle gradients
    m2 = fast_cpu.CpuBLSTMCTC(sd, L, cell_clip=50., clip_grad_norm=5.0, optimizer='rmsprop')
    m2.train_step(x, labs, sl, lr=1e-3)
    for k in sd:
        g = oopt.clip_by_norm(r['grads'][k], 5.0)
        s0, s1 = oopt.init_slots('rmsprop', np.asarray(sd[k], np.float64))
        want, _, _ = oopt.step('rmsprop', np.asarray(sd[k], np.float64), g, s0, s1, 1e-3, 1)
        assert np.abs(m2.params[k].detach().numpy() - want).max() < 1e-5, k


def test_sigmoid_smoothing_backward_formula():
    """The closed form att_softmax_bwd_kernel uses for sigmoid smoothing,
    de = k * alpha * (1 - alpha * S) * (da - sum_j alpha_j da_j), S = sum_t sigmoid(k e_t),
    against autograd through the oracle's normalisation (oracle/attention.py attention_step)."""
    import torch
    rng = np.random.RandomState(4)
    B, T, k = 3, 11, 1.7
    e = torch.tensor(rng.randn(B, T) * 2.0, dtype=torch.float64, requires_grad=True)
    da = torch.tensor(rng.randn(B, T), dtype=torch.float64)
    sg = torch.sigmoid(e * k)
    S = sg.sum(1, keepdim=True)
    alpha = sg / S
    (alpha * da).sum().backward()
    a = alpha.detach()
    dot = (a * da).sum(1, keepdim=True)
    closed = k * a * (1.0 - a * S.detach()) * (da - dot)
    assert (closed - e.grad).abs().max().item() < 1e-12


def test_ctc_matches_tensorflow_known_answer_vectors():
    """oracle/ctc.py against the constants of TensorFlow's own ctc_loss_op_test.py testBasic
    (tests/golden/tf_known_answers.py): both losses and all 60 gradient entries."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import ctc as octc
    for probs, targets, loss, grad in tfk.CTC_CASES:
        l, g, ok = octc.ctc_loss_single(np.log(probs), targets)           # blank = depth - 1
        assert ok and abs(l - loss) < 5e-6
        assert np.abs(g - grad).max() < 2e-6
    # batched, time-major, padded to 7 frames as in that test
    logits = np.zeros((7, 2, tfk.CTC_DEPTH))
    for b, (probs, _, _, _) in enumerate(tfk.CTC_CASES):
        logits[:5, b] = np.log(probs)
    loss, grad = octc.ctc_loss_batch(logits, [c[1] for c in tfk.CTC_CASES], [5, 5])
    assert np.abs(loss - [tfk.CTC_LOSS_0, tfk.CTC_LOSS_1]).max() < 5e-6
    assert np.abs(grad[:5, 0] - tfk.CTC_GRAD_0).max() < 2e-6 and not grad[5:].any()


def test_lstm_cell_matches_tensorflow_known_answer():
    """oracle/lstm.py lstm_block_cell (no peephole, forget_bias 1) against the constants of TensorFlow's own
    lstm_ops_test.py testLSTMBlockCell: two stacked 2-unit cells, all weights 0.5."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import lstm as olstm
    H = 2
    z = torch.zeros(H, dtype=torch.float64)
    w = torch.full((2 + H, 4 * H), tfk.LSTM_WEIGHT, dtype=torch.float64)
    b = torch.zeros(4 * H, dtype=torch.float64)
    st = torch.full((1, H), tfk.LSTM_STATE, dtype=torch.float64)
    x = torch.tensor([tfk.LSTM_X], dtype=torch.float64)
    c0, h0 = olstm.lstm_block_cell(x, st, st, w, b, z, z, z, forget_bias=1.0, cell_clip=0.0, use_peephole=False)
    c1, h1 = olstm.lstm_block_cell(h0, st, st, w, b, z, z, z, forget_bias=1.0, cell_clip=0.0, use_peephole=False)
    for got, want in ((c0, tfk.LSTM_C0), (h0, tfk.LSTM_H0), (c1, tfk.LSTM_C1), (h1, tfk.LSTM_H1)):
        assert np.abs(got.numpy()[0] - np.asarray(want)).max() < 2e-7      # TensorFlow printed float32 values


def test_greedy_decoder_matches_tensorflow_known_answer():
    """oracle/decoders.py greedy_decode against TensorFlow's own ctc_decoder_ops_test.py testCTCGreedyDecoder
    (best path, repeats merged, blank = depth - 1; zero-probability entries are -inf log-probabilities)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import decoders as odec
    with np.errstate(divide='ignore'):
        logp = np.log(tfk.GREEDY_PROBS)
    got = odec.greedy_decode(logp, tfk.GREEDY_SEQ_LEN, 3)
    assert [list(map(int, g)) for g in got] == tfk.GREEDY_DECODED
    for b, n in enumerate(tfk.GREEDY_SEQ_LEN):
        assert abs(float(-logp[b, :n].max(1).sum()) - tfk.GREEDY_NEG_LOG_PROB[b]) < 1e-12


def test_beam_search_decoder_matches_tensorflow_known_answer():
    """oracle/decoders.py beam_search_decode against TensorFlow's own ctc_decoder_ops_test.py testCTCDecoderBeamSearch:
    beam_width 2, top_paths 2 -> beams [1, 0] and [0, 1, 0]; TF1's log_probability is log p(path) plus the frame-max
    normaliser (it divides every frame by its maximum, not by its sum), reproduced to the constants' six digits.  The
    logit offset (+2.0) and the frames beyond seq_len must not matter."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import decoders as odec
    logits = np.zeros((1, tfk.BEAM_PADDED_FRAMES, tfk.BEAM_PROBS.shape[1]))
    logits[0, :tfk.BEAM_PROBS.shape[0]] = np.log(tfk.BEAM_PROBS) + tfk.BEAM_LOGIT_OFFSET
    lp = logits - np.log(np.exp(logits).sum(2, keepdims=True))            # the decoders take log-softmax
    paths, nll = odec.beam_search_decode(lp, [tfk.BEAM_SEQ_LEN], tfk.BEAM_BLANK, beam_width=tfk.BEAM_WIDTH, top_paths=2)
    assert paths[0] == tfk.BEAM_DECODED
    norm = tfk.beam_max_normaliser(tfk.BEAM_PROBS, tfk.BEAM_SEQ_LEN)
    assert np.abs(-nll[0] + norm - np.asarray(tfk.BEAM_LOG_PROB)).max() < 2e-6
    best, nll1 = odec.beam_search_decode(lp, [tfk.BEAM_SEQ_LEN], tfk.BEAM_BLANK, beam_width=tfk.BEAM_WIDTH)
    assert best[0] == tfk.BEAM_DECODED[0] and abs(nll1[0] - nll[0][0]) < 1e-12
    # merge_repeated=True (the reference's call) collapses repeats of an output beam; these two have none
    assert [odec.merge_repeated(p) for p in paths[0]] == tfk.BEAM_DECODED and odec.merge_repeated([3, 3, 1, 1, 3]) == [3, 1, 3]
    # ... and a path WITH repeats: the worked example of TensorFlow's own documentation of the op (A B B * B * B)
    lp = np.log(tfk.merge_doc_probs())[None]
    for width in (2, 4, 8):
        best, _ = odec.beam_search_decode(lp, [lp.shape[1]], tfk.MERGE_DOC_BLANK, beam_width=width)
        assert best[0] == tfk.MERGE_DOC_UNMERGED and odec.merge_repeated(best[0]) == tfk.MERGE_DOC_MERGED


def test_adagrad_matches_tensorflow_known_answer():
    """oracle/optim.py 'adagrad' (accumulator starts at 0.1, TF1 default) against the constants of TensorFlow's
    own adagrad_test.py doTestBasic: three steps at learning rate 3.0."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import optim as oopt
    for var, grad, want in ((tfk.ADAGRAD_VAR0, tfk.ADAGRAD_GRAD0, tfk.ADAGRAD_OUT0),
                            (tfk.ADAGRAD_VAR1, tfk.ADAGRAD_GRAD1, tfk.ADAGRAD_OUT1)):
        p, g = np.asarray(var, dtype=np.float64), np.asarray(grad, dtype=np.float64)
        s0, s1 = oopt.init_slots('adagrad', p)
        for t in range(1, tfk.ADAGRAD_STEPS + 1):
            p, s0, s1 = oopt.step('adagrad', p, g, s0, s1, tfk.ADAGRAD_LR, t)
        assert np.abs(p - want).max() < 5e-7          # TensorFlow's constants are float32 results


def test_sgd_momentum_nesterov_rmsprop_adam_match_tensorflow_known_answers():
    """oracle/optim.py update rules against what TensorFlow's own optimizer tests expect (gradient_descent_test.py,
    momentum_test.py testBasic + the numpy reference of testNesterovMomentum, rmsprop_test.py testWithoutMomentum --
    which is what fixes "rms slot starts at 1" and "epsilon inside the root" --, adam_test.py's adam_update_numpy)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import optim as oopt
    A = lambda v: np.asarray(v, dtype=np.float64)
    for i in range(2):
        # sgd
        p, s0, s1 = oopt.step('sgd', A(tfk.SGD_VAR[i]), A(tfk.SGD_GRAD[i]), None, None, tfk.SGD_LR, 1)
        assert np.abs(p - A(tfk.SGD_OUT[i])).max() < 1e-15
        # momentum: two steps, variables and accumulator
        p, g = A(tfk.MOM_VAR[i]), A(tfk.MOM_GRAD[i])
        s0, s1 = oopt.init_slots('momentum', p)
        assert np.all(s0 == 0)
        p, s0, s1 = oopt.step('momentum', p, g, s0, s1, tfk.MOM_LR, 1, momentum=tfk.MOM_MOMENTUM)
        assert np.abs(p - A(tfk.MOM_OUT_STEP1[i])).max() < 1e-15
        p, s0, s1 = oopt.step('momentum', p, g, s0, s1, tfk.MOM_LR, 2, momentum=tfk.MOM_MOMENTUM)
        assert np.abs(p - A(tfk.MOM_OUT_STEP2[i])).max() < 1e-15 and np.abs(s0 - A(tfk.MOM_ACCUM_STEP2[i])).max() < 1e-15
        # nesterov: TensorFlow's numpy reference, changing gradients, five steps
        p = q = A(tfk.MOM_VAR[i])
        s0, _ = oopt.init_slots('nestrov', p)
        acc = np.zeros_like(q)
        for t in range(1, 6):
            g = A(tfk.MOM_GRAD[i]) * (1.0 + 0.3 * t) * (-1) ** t
            p, s0, _ = oopt.step('nestrov', p, g, s0, None, 2.0, t, momentum=0.9)
            q, acc = tfk.nesterov_reference(q, acc, g, 2.0, 0.9)
            assert np.abs(p - q).max() < 1e-14 and np.abs(s0 - acc).max() < 1e-14
        # rmsprop: slot initial value, epsilon placement, two steps
        p, g = A(tfk.RMS_VAR[i]), A(tfk.RMS_GRAD[i])
        s0, s1 = oopt.init_slots('rmsprop', p)
        assert np.all(s0 == 1.0)
        p, s0, s1 = oopt.step('rmsprop', p, g, s0, s1, tfk.RMS_LR, 1, decay=tfk.RMS_DECAY, rms_eps=tfk.RMS_EPS)
        assert np.abs(s0 - A(tfk.RMS_SLOT_STEP1[i])).max() < 1e-15 and np.abs(p - A(tfk.RMS_OUT_STEP1[i])).max() < 1e-15
        p, s0, s1 = oopt.step('rmsprop', p, g, s0, s1, tfk.RMS_LR, 2, decay=tfk.RMS_DECAY, rms_eps=tfk.RMS_EPS)
        assert np.abs(s0 - A(tfk.RMS_SLOT_STEP2[i])).max() < 1e-15 and np.abs(p - A(tfk.RMS_OUT_STEP2[i])).max() < 1e-14
        # adam: three steps against adam_update_numpy
        p = q = A(tfk.ADAM_VAR[i])
        g = A(tfk.ADAM_GRAD[i])
        s0, s1 = oopt.init_slots('adam', p)
        m = v = np.zeros_like(q)
        for t in range(1, tfk.ADAM_STEPS + 1):
            p, s0, s1 = oopt.step('adam', p, g, s0, s1, 0.001, t)
            q, m, v = tfk.adam_reference(q, g, t, m, v)
            assert np.abs(p - q).max() < 1e-15
        # ... which for a constant gradient moves every entry by lr * g / (|g| + eps / sqrt(1 - beta2^t)) per step
        want = A(tfk.ADAM_VAR[i]) - sum(0.001 * g / (np.abs(g) + 1e-8 / np.sqrt(1 - 0.999 ** t)) for t in (1, 2, 3))
        assert np.abs(p - want).max() < 1e-12


def test_adadelta_matches_tensorflow_known_answer():
    """oracle/optim.py's Adadelta against the scalar recurrence TensorFlow's adadelta_test.py::doTestBasic carries and
    asserts step by step (both slots and both variables, 3 gradients x 3 learning rates, 4 updates; slots start at 0,
    epsilon inside both roots, the learning rate scales the update only)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import optim as oopt
    for grad in tfk.ADADELTA_GRADS:
        for lr in tfk.ADADELTA_LRS:
            want = tfk.adadelta_reference(grad, lr, tfk.ADADELTA_STEPS, tfk.ADADELTA_RHO, tfk.ADADELTA_EPS)
            for var in tfk.ADADELTA_VAR:
                p0 = np.asarray(var, dtype=np.float64)
                p, g = p0.copy(), np.full(2, grad)
                s0, s1 = oopt.init_slots('adadelta', p)
                assert np.all(s0 == 0) and np.all(s1 == 0)
                for t in range(1, tfk.ADADELTA_STEPS + 1):
                    p, s0, s1 = oopt.step('adadelta', p, g, s0, s1, lr, t)
                    accum, accum_update, tot = want[t - 1]
                    assert np.abs(s0 - accum).max() < 1e-15 and np.abs(s1 - accum_update).max() < 1e-15
                    assert np.abs(p - (p0 - tot)).max() < 1e-14


def test_clip_by_norm_matches_tensorflow_known_answer():
    """oracle/optim.py clip_by_norm against TensorFlow's clip_ops_test.py testClipByNormClipped / NotClipped."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import optim as oopt
    x = np.asarray(tfk.CLIP_X)
    assert np.abs(oopt.clip_by_norm(x, tfk.CLIP_NORM_CLIPPED) - np.asarray(tfk.CLIP_ANS_CLIPPED)).max() < 1e-12
    assert np.array_equal(oopt.clip_by_norm(x, tfk.CLIP_NORM_NOT_CLIPPED), x)


def test_conv_and_pool_conventions_match_tensorflow_known_answers():
    """oracle/vgg.py: the NHWC / HWIO -> NCHW / OIHW permutes of its convolutions and the pad-AFTER rule of its SAME
    max-pool, against the constants of TensorFlow's conv_ops_test.py (testConv2D2x2Filter, testConv2D1x2Filter) and
    pooling_ops_test.py (testMaxPoolSamePadding)."""
    import sys
    import torch
    import torch.nn.functional as Fn
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import vgg as ovgg
    n_in = int(np.prod(tfk.CONV_IN_SHAPE))
    x = torch.arange(1, n_in + 1, dtype=torch.float64).reshape(tfk.CONV_IN_SHAPE)            # NHWC
    for fshape, want in ((tfk.CONV_2X2_FILTER_SHAPE, tfk.CONV_2X2_OUT), (tfk.CONV_1X2_FILTER_SHAPE, tfk.CONV_1X2_OUT)):
        w = torch.arange(1, int(np.prod(fshape)) + 1, dtype=torch.float64).reshape(fshape)   # HWIO
        y = Fn.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1))                          # the oracle's permutes
        assert y.permute(0, 2, 3, 1).reshape(-1).tolist() == want
    # ... and oracle._conv is exactly that expression with SAME padding 1, bias and ReLU
    rng = np.random.RandomState(0)
    xi = torch.tensor(rng.randn(2, 3, 5, 4))
    w3, b3 = torch.tensor(rng.randn(3, 3, 3, 6)), torch.tensor(rng.randn(6))
    want = torch.relu(Fn.conv2d(Fn.pad(xi, (1, 1, 1, 1)), w3.permute(3, 2, 0, 1)) + b3.view(1, -1, 1, 1))
    assert (ovgg._conv(xi, w3, b3) - want).abs().max().item() < 1e-12
    pooled = ovgg._pool_same(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).reshape(-1).tolist()
    assert pooled == tfk.POOL_SAME_OUT


@pytest.mark.parametrize('reverse', [False, True])
@pytest.mark.parametrize('clip,peep', [(0.0, True), (0.6, True), (0.6, False)])
def test_lstm_explicit_bptt_matches_autograd(reverse, clip, peep):
    """Second implementation of the peephole / cell-clip / sequence_length rules: the numpy forward +
    hand-derived BPTT (oracle.lstm.layer_*_np, LSTMBlockCellGrad form with the straight-through clip)
    against the torch-autograd statement, ragged lengths incl. 0 and 1, gradients entering through
    the outputs and the final (c, h)."""
    rng = np.random.RandomState(3 + int(reverse))
    T, B, D, H = 11, 6, 5, 8
    lens = np.array([11, 3, 1, 0, 7, 11])
    x = rng.randn(B, T, D) * 2
    for b in range(B):
        x[b, lens[b]:] = 0
    p = olstm.init_lstm_params(rng, D, H, 0.5)
    p['b'] = torch.tensor(rng.randn(4 * H) * 0.3)
    pt = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xt = torch.tensor(x).transpose(0, 1).contiguous().requires_grad_(True)
    out, (cf, hf) = olstm.dynamic_rnn(xt, torch.tensor(lens), pt, reverse=reverse, cell_clip=clip, use_peephole=peep)
    dout, dcf, dhf = rng.randn(T, B, H), rng.randn(B, H), rng.randn(B, H)
    ((out * torch.tensor(dout)).sum() + (cf * torch.tensor(dcf)).sum() + (hf * torch.tensor(dhf)).sum()).backward()
    pn = {k: v.numpy() for k, v in p.items()}
    f = olstm.layer_forward_np(xt.detach().numpy(), lens, pn, reverse, 1.0, clip, peep)
    if clip:
        assert np.abs(f['cs']).max() == clip      # the clip is active in this case
    assert np.abs(f['hout'] - out.detach().numpy()).max() < 1e-13
    assert np.abs(f['c_final'] - cf.detach().numpy()).max() < 1e-13
    assert np.abs(f['h_final'] - hf.detach().numpy()).max() < 1e-13
    bw = olstm.layer_backward_np(dout, f['gates'], f['cs'], lens, pn, reverse, peep, dcf, dhf)
    dw, dx = olstm.layer_param_grads_np(xt.detach().numpy(), f['hout'], bw['dgates'], lens, pn, reverse)
    assert np.abs(dw - pt['w'].grad.numpy()).max() < 1e-12
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-12
    assert np.abs(bw['db'] - pt['b'].grad.numpy()).max() < 1e-12
    if peep:
        for k, n in enumerate(('wci', 'wcf', 'wco')):
            assert np.abs(bw['dpeep'][k] - pt[n].grad.numpy()).max() < 1e-12
    # the rounded variants of the two statements agree with each other as well (bf16 operand points)
    xr = olstm.bf16_round(xt.detach().numpy())
    pr = {k: v.clone() for k, v in p.items()}
    pr['w'] = olstm.bf16_round_t(pr['w'])
    o2, _ = olstm.dynamic_rnn(torch.tensor(xr), torch.tensor(lens), pr, reverse=reverse, cell_clip=clip,
                              use_peephole=peep, h_round=olstm.bf16_round_t)
    f2 = olstm.layer_forward_np(xr, lens, pn, reverse, 1.0, clip, peep, round_fn=olstm.bf16_round)
    assert np.abs(f2['hout'] - o2.numpy()).max() < 1e-12


def test_gru_cell_matches_tensorflow_known_answer():
    """oracle.gru.gru_cell against the constants of TensorFlow's core_rnn_cell_test.py::testGRUCell: pins the GRUCell
    equations (reset gate before the candidate's matrix product, h' = u h + (1-u) c, gate bias 1, candidate bias 0)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf_known_answers as tfk
    from oracle import gru as ogru
    for case in tfk.GRU_CASES:
        D, H = len(case['x']), len(case['h'])
        p = dict(wg=torch.full((D + H, 2 * H), case['kernel'], dtype=torch.float64), bg=torch.ones(2 * H, dtype=torch.float64),
                 wc=torch.full((D + H, H), case['kernel'], dtype=torch.float64), bc=torch.zeros(H, dtype=torch.float64))
        out = ogru.gru_cell(torch.tensor([case['x']], dtype=torch.float64), torch.tensor([case['h']], dtype=torch.float64), p)
        assert np.abs(out.numpy()[0] - np.array(case['out'])).max() < 1e-6


def test_gru_dynamic_rnn_masking_and_reverse():
    """sequence_length handling of the GRU oracle: outputs past the length are zero, the final state is the state at
    the last valid frame, and the backward direction of an utterance equals the forward direction of its reversal."""
    from oracle import gru as ogru
    rng = np.random.RandomState(3)
    T, B, D, H = 7, 3, 4, 5
    p = dict(wg=torch.tensor(rng.randn(D + H, 2 * H) * 0.4), bg=torch.tensor(rng.randn(2 * H) * 0.1 + 1),
             wc=torch.tensor(rng.randn(D + H, H) * 0.4), bc=torch.tensor(rng.randn(H) * 0.1))
    x = torch.tensor(rng.randn(T, B, D))
    sl = torch.tensor([7, 4, 2])
    out, hf = ogru.dynamic_rnn(x, sl, p)
    for b in range(B):
        n = int(sl[b])
        assert float(out[n:, b].abs().sum()) == 0.0
        assert torch.allclose(hf[b], out[n - 1, b])
        alone, hfa = ogru.dynamic_rnn(x[:n, b:b + 1], torch.tensor([n]), p)
        assert torch.allclose(alone[:, 0], out[:n, b], atol=1e-12)
        rev, hfr = ogru.dynamic_rnn(x[:, b:b + 1], sl[b:b + 1], p, reverse=True)
        flipped, hff = ogru.dynamic_rnn(torch.flip(x[:n, b:b + 1], [0]), torch.tensor([n]), p)
        assert torch.allclose(rev[:n, 0], torch.flip(flipped[:, 0], [0]), atol=1e-12) and torch.allclose(hfr, hff, atol=1e-12)

