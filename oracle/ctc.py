"""Oracle (test infrastructure; PINNED to TensorFlow's own known-answer vectors of ctc_loss_op_test.py testBasic,
tests/golden/tf_known_answers.py -- see oracle/__init__.py): CPU restatement of tf.nn.ctc_loss as the reference
calls it.

Call sites restated: models/ctc/ctc.py:289-298 (ctc_merge_repeated=True,
preprocess_collapse_repeated=False, ignore_longer_outputs_than_inputs=True,
time_major=True, mean over the batch) and
models/attention/joint_ctc_attention.py:308-317 (same, ignore_longer=False).

Algorithm (Graves 2006 as cited in README.md:48):
blank = C-1; y = softmax(logits); l' = blank-interleaved labels, S = 2L+1;
  alpha_0(0) = ln y_0(blank), alpha_0(1) = ln y_0(l_1)
  alpha_t(s) = ln y_t(l'_s) + LSE(alpha_{t-1}(s), alpha_{t-1}(s-1),
                                  [alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2}])
  loss = -LSE(alpha_{T-1}(S-1), alpha_{T-1}(S-2))
  d loss / d logit_t(k) = y_t(k) - sum_{s: l'_s = k} exp(alpha_t(s) + beta_t(s) - ln y_t(l'_s) + loss)
(beta defined WITH the emission at t, the textbook form; TF's kernel keeps beta
without it -- same product.)  Frames t >= seq_len[b] get zero gradient.
Infeasible utterances (no valid alignment) give loss 0 / grad 0 under
ignore_longer_outputs_than_inputs=True.

Independent cross-check: torch.nn.functional.ctc_loss (tests/test_oracle.py).
"""
import numpy as np

NEG_INF = -np.inf


def _lse2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid='ignore', divide='ignore'):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def log_softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = x - m
    return e - np.log(np.exp(e).sum(axis=axis, keepdims=True))


def ctc_loss_single(logits_tc, labels, blank=None):
    """logits_tc [T,C] (already cut to seq_len), labels 1-D int.  Returns
    (loss, grad [T,C], feasible)."""
    logits_tc = np.asarray(logits_tc, dtype=np.float64)
    T, C = logits_tc.shape
    if blank is None:
        blank = C - 1
    lab = np.asarray(labels, dtype=np.int64)
    L = len(lab)
    S = 2 * L + 1
    ext = np.full(S, blank, dtype=np.int64)
    ext[1::2] = lab
    logp = log_softmax(logits_tc)
    lp = logp[:, ext]                                    # [T,S]
    skip = np.zeros(S, dtype=bool)
    skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])

    alpha = np.full((T, S), NEG_INF)
    if T > 0:
        alpha[0, 0] = lp[0, 0]
        if S > 1:
            alpha[0, 1] = lp[0, 1]
    for t in range(1, T):
        a = alpha[t - 1]
        acc = a.copy()
        acc[1:] = _lse2(acc[1:], a[:-1])
        tmp = np.full(S, NEG_INF)
        tmp[2:] = np.where(skip[2:], a[:-2], NEG_INF)
        acc = _lse2(acc, tmp)
        alpha[t] = acc + lp[t]
    if T == 0:
        return 0.0, np.zeros_like(logits_tc), False
    ll = alpha[T - 1, S - 1]
    if S > 1:
        ll = _lse2(ll, alpha[T - 1, S - 2])
    ll = float(ll)
    if not np.isfinite(ll):
        return 0.0, np.zeros_like(logits_tc), False

    beta = np.full((T, S), NEG_INF)
    beta[T - 1, S - 1] = lp[T - 1, S - 1]
    if S > 1:
        beta[T - 1, S - 2] = lp[T - 1, S - 2]
    for t in range(T - 2, -1, -1):
        b = beta[t + 1]
        acc = b.copy()
        acc[:-1] = _lse2(acc[:-1], b[1:])
        tmp = np.full(S, NEG_INF)
        tmp[:-2] = np.where(skip[2:], b[2:], NEG_INF)
        acc = _lse2(acc, tmp)
        beta[t] = acc + lp[t]

    with np.errstate(invalid='ignore'):
        gamma = np.exp(alpha + beta - lp - ll)           # posterior occupation [T,S]
    gamma = np.where(np.isfinite(gamma), gamma, 0.0)
    grad = np.exp(logp)
    for s in range(S):
        grad[:, ext[s]] -= gamma[:, s]
    return -ll, grad, True


def ctc_loss_batch(logits_tbc, labels_list, seq_len, blank=None, ignore_longer=True):
    """tf.nn.ctc_loss(time_major=True).  logits [T,B,C]; labels_list: list of B
    int sequences; seq_len [B].  Returns loss [B], grad [T,B,C] (d loss_b/d logits)."""
    T, B, C = logits_tbc.shape
    loss = np.zeros(B)
    grad = np.zeros((T, B, C))
    for b in range(B):
        n = int(seq_len[b])
        l, g, ok = ctc_loss_single(logits_tbc[:n, b], labels_list[b], blank)
        if not ok and not ignore_longer:
            raise ValueError('Not enough time for target transition sequence '
                             '(utterance %d)' % b)
        loss[b] = l
        grad[:n, b] = g
    return loss, grad


def dense_to_list(labels_dense, pad=-1):
    """Rows of a [B,Lmax] array padded with `pad`; scanning stops at the first
    pad (utils/io/labels/sparsetensor.py:26-32)."""
    out = []
    for row in np.asarray(labels_dense):
        seq = []
        for v in row:
            if v == pad:
                break
            seq.append(int(v))
        out.append(seq)
    return out


def list2sparsetensor(labels, padded_value):
    """Restatement of utils/io/labels/sparsetensor.py:12-39 (that module imports
    tensorflow at top so cannot be imported here)."""
    indices, values = [], []
    for i_utt, each in enumerate(labels):
        for i_l, l in enumerate(each):
            if l == padded_value:
                break
            indices.append([i_utt, i_l])
            values.append(l)
    dense_shape = [len(labels), np.asarray(indices).max(0)[1] + 1]
    return [np.array(indices, dtype=np.int64), np.array(values, dtype=np.int32),
            np.array(dense_shape, dtype=np.int64)]


def sparse_to_list(st, batch_size):
    """Per-utterance label lists from the (indices, values, shape) triple."""
    out = [[] for _ in range(batch_size)]
    for (b, _), v in zip(st[0], st[1]):
        out[int(b)].append(int(v))
    return out


def edit_distance(hyp, ref):
    """Levenshtein distance (tf.edit_distance / python-Levenshtein restated)."""
    n, m = len(hyp), len(ref)
    prev = list(range(m + 1))
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        for j in range(1, m + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (hyp[i - 1] != ref[j - 1]))
        prev = cur
    return prev[m]


def label_error_rate(hyps, refs):
    """tf.reduce_mean(tf.edit_distance(hyp, truth, normalize=True)) (ctc.py:391)."""
    vals = []
    for h, r in zip(hyps, refs):
        d = edit_distance(list(h), list(r))
        vals.append(d / len(r) if len(r) > 0 else (0.0 if d == 0 else np.inf))
    return float(np.mean(vals))
