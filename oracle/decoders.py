"""Oracle (test infrastructure, PINNED by tests/golden/decoders_*.npz which were
produced by the reference's own numpy decoders -- tests/golden/make_golden.py):
CPU restatement of the CTC decoders.

  * greedy: models/ctc/decoders/greedy_decoder.py:19-50 (per-frame argmax over
    log(probs), itertools.groupby collapse, drop blank) == tf.nn.ctc_greedy_decoder
    (models/ctc/ctc.py:341-342; first max wins on ties, blank = C-1).
  * beam: models/ctc/decoders/beam_search_decoder.py:53-152 (prefix beam
    search in log space).  Ordering rules that decide ties, kept exactly:
      - candidates are inserted into next_beam in vocab-major order
        (for c in classes: for prefix in beam) -> dict insertion order,
      - sorted(..., key=logsumexp(p_b,p_nb), reverse=True) is a STABLE sort
        (equal scores keep insertion order), then truncated to beam_width.
    alpha/beta (LM weight, insertion bonus) are accepted and unused, as in
    the reference (:53,:129).
"""
import math
import numpy as np

NEG_INF = -float('inf')


def greedy_decode(log_probs_btc, seq_len, blank):
    """log_probs (or logits: argmax is the same) [B,T,C] -> list of lists."""
    out = []
    for b in range(log_probs_btc.shape[0]):
        n = int(seq_len[b])
        idx = np.argmax(log_probs_btc[b, :n], axis=1) if n > 0 else np.zeros(0, np.int64)
        hyp = []
        prev = None
        for a in idx:
            a = int(a)
            if a != prev:
                if a != blank:
                    hyp.append(a)
            prev = a
        out.append(hyp)
    return out


def _lse(*args):
    """beam_search_decoder.py:23-32."""
    if all(a == NEG_INF for a in args):
        return NEG_INF
    m = max(args)
    return m + math.log(sum(math.exp(a - m) for a in args))


def beam_search_decode(log_probs_btc, seq_len, blank, beam_width=1, top_paths=None):
    """Returns (list of best prefixes, np.array of -log scores); with top_paths = k the first k beams of every
    utterance instead: (list of lists of prefixes, [B, k] -log scores) -- tf.nn.ctc_beam_search_decoder(top_paths=k,
    merge_repeated=False); pinned to TensorFlow's own test constants in tests/golden/tf_known_answers.py."""
    B, T, C = log_probs_btc.shape
    results, scores = [], []
    for b in range(B):
        beam = [(tuple(), (0.0, NEG_INF))]
        for t in range(int(seq_len[b])):
            nxt = {}
            lp_t = log_probs_btc[b, t]
            for c in range(C):
                p_t = float(lp_t[c])
                for prefix, (p_b, p_nb) in beam:
                    if c == blank:
                        nb, nnb = nxt.get(prefix, (NEG_INF, NEG_INF))
                        nxt[prefix] = (_lse(nb, p_b + p_t, p_nb + p_t), nnb)
                        continue
                    end = prefix[-1] if prefix else None
                    new_prefix = prefix + (c,)
                    nb, nnb = nxt.get(new_prefix, (NEG_INF, NEG_INF))
                    if c != end:
                        nnb = _lse(nnb, p_b + p_t, p_nb + p_t)
                    else:
                        nnb = _lse(nnb, p_b + p_t)
                    nxt[new_prefix] = (nb, nnb)
                    if c == end:
                        nb, nnb = nxt.get(prefix, (NEG_INF, NEG_INF))
                        nxt[prefix] = (nb, _lse(nnb, p_nb + p_t))
            beam = sorted(nxt.items(), key=lambda kv: _lse(*kv[1]), reverse=True)[:beam_width]
        if top_paths is None:
            results.append(list(beam[0][0]))
            scores.append(-_lse(*beam[0][1]))
        else:
            results.append([list(e[0]) for e in beam[:top_paths]])
            scores.append([-_lse(*e[1]) for e in beam[:top_paths]])
    return results, np.array(scores)


def merge_repeated(path):
    """tf.nn.ctc_beam_search_decoder(merge_repeated=True), the reference's call (models/ctc/ctc.py:344-346): consecutive
    equal labels of an OUTPUT beam collapse to one."""
    return [c for i, c in enumerate(path) if i == 0 or c != path[i - 1]]
