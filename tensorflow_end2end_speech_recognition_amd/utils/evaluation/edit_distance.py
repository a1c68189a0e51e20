"""Edit distance for LER -- stands in for tf.edit_distance(normalize=True)
(models/ctc/ctc.py:391) and python-Levenshtein in utils/evaluation/edit_distance.py:35-71."""
import numpy as np


def levenshtein(hyp, ref):
    hyp, ref = list(hyp), list(ref)
    n, m = len(hyp), len(ref)
    if m == 0:
        return n
    prev = np.arange(m + 1)
    for i in range(1, n + 1):
        cur = np.empty(m + 1, dtype=np.int64)
        cur[0] = i
        h = hyp[i - 1]
        for j in range(1, m + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (h != ref[j - 1]))
        prev = cur
    return int(prev[m])


def compute_ler(hyps, refs):
    """mean_b edit_distance(hyp_b, ref_b) / len(ref_b)."""
    vals = []
    for h, r in zip(hyps, refs):
        d = levenshtein(h, r)
        vals.append(d / len(r) if len(r) > 0 else (0.0 if d == 0 else float('inf')))
    return float(np.mean(vals)) if vals else 0.0


# ---- utils/evaluation/edit_distance.py:35-109: PER / CER / WER on host sequences.  The reference calls
# python-Levenshtein's `distance` (unit-cost edit distance); `levenshtein` above is that function.
def compute_per(ref, hyp, normalize=True):
    """Phone error rate between two lists of phone strings; divided by len(ref) (:35-56)."""
    per = levenshtein(list(ref), list(hyp))
    return per / len(ref) if normalize else per


def compute_cer(str_pred, str_true, normalize=True):
    """Character error rate between two strings without spaces; divided by len(str_true) (:59-71)."""
    cer = levenshtein(list(str_pred), list(str_true))
    return cer / len(list(str_true)) if normalize else cer


def _wer_table(ref, hyp):
    d = np.zeros((len(ref) + 1, len(hyp) + 1), dtype=np.int64)
    d[0, :] = np.arange(len(hyp) + 1)
    d[:, 0] = np.arange(len(ref) + 1)
    for i in range(1, len(ref) + 1):
        for j in range(1, len(hyp) + 1):
            if ref[i - 1] == hyp[j - 1]:
                d[i, j] = d[i - 1, j - 1]
            else:
                d[i, j] = min(d[i - 1, j - 1], d[i, j - 1], d[i - 1, j]) + 1
    return d


def compute_wer(ref, hyp, normalize=True):
    """Word error rate between two word lists; divided by len(ref) (:74-109)."""
    wer = _wer_table(ref, hyp)[len(ref), len(hyp)]
    return wer / len(ref) if normalize else wer


def wer_align(ref, hyp):
    """(substitutions, insertions, deletions) of one optimal alignment, back-traced with the reference's
    preference order match > insertion > substitution > deletion (:112-280; the aligned print-out is not
    reproduced)."""
    d = _wer_table(ref, hyp)
    x, y = len(ref), len(hyp)
    sub = ins = dele = 0
    while x > 0 or y > 0:
        if x > 0 and y > 0 and d[x, y] == d[x - 1, y - 1] and ref[x - 1] == hyp[y - 1]:
            x, y = x - 1, y - 1
        elif y > 0 and d[x, y] == d[x, y - 1] + 1:
            ins += 1
            y -= 1
        elif x > 0 and y > 0 and d[x, y] == d[x - 1, y - 1] + 1:
            sub += 1
            x, y = x - 1, y - 1
        else:
            dele += 1
            x -= 1
    return sub, ins, dele


def tf_edit_distance(hypothesis_st, truth_st, normalize=True):
    """tf.edit_distance(hypothesis, truth, normalize) on two [indices, values, dense_shape] triples: per batch row the
    Levenshtein distance, divided by len(truth) when normalize (inf for an empty truth against a non-empty hypothesis,
    0 when both are empty) -- the cases of the op's documentation are pinned in tests/test_host_io.py."""
    import numpy as np
    from ..io.labels.sparsetensor import sparse_to_flat
    B = int(np.asarray(truth_st[2])[0])
    hv, ho, _ = sparse_to_flat(hypothesis_st, B)
    tv, to, _ = sparse_to_flat(truth_st, B)
    out = np.zeros(B, dtype=np.float64)
    for b in range(B):
        hyp, truth = hv[ho[b]:ho[b + 1]], tv[to[b]:to[b + 1]]
        d = levenshtein(list(hyp), list(truth))
        out[b] = float(d) if not normalize else (d / len(truth) if len(truth) else (float('inf') if d else 0.0))
    return out


def compute_edit_distance(session, labels_true_st, labels_pred_st):
    """What the reference's compute_edit_distance (utils/evaluation/edit_distance.py:15-32) returns -- including its
    argument swap: it builds `labels_pred_pl` from labels_TRUE_st and `labels_true_pl` from labels_PRED_st and calls
    tf.edit_distance(labels_pred_pl, labels_true_pl, normalize=True), so the transcript is passed as the hypothesis
    and the PREDICTION as the truth the distance is divided by: distance / len(prediction), not / len(transcript)
    (the Levenshtein distance itself is symmetric).  `session` is accepted for call compatibility and ignored.
    CTC.compute_ler (models/ctc/ctc.py:391) calls tf.edit_distance(decode_op, labels) directly and does divide by the
    transcript length -- compute_ler above."""
    return tf_edit_distance(labels_true_st, labels_pred_st, normalize=True)
