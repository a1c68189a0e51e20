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
