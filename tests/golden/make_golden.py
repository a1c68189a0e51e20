#!/usr/bin/env python
"""Generate golden vectors from the REFERENCE'S OWN code (run in the build
container only: needs /root/reference; the .npz outputs are committed and are
what travels to the GPU box).

    python tests/golden/make_golden.py

Produces
  decoders_v1.npz : seeded posteriors + outputs of
      /root/reference/models/ctc/decoders/greedy_decoder.py  GreedyDecoder
      /root/reference/models/ctc/decoders/beam_search_decoder.py BeamSearchDecoder
    called with B=1 slices (as the reference itself does at
    examples/librispeech/metrics/ctc.py:220-223; B>1 ragged np.array raises on
    numpy >= 1.24).  Posteriors are float64 so the reference arithmetic is
    float64 under both old (value-based) and NEP-50 numpy promotion rules.
  splice_v1.npz   : inputs/outputs of utils/io/inputs/splicing.py do_splice and
      utils/io/inputs/frame_stacking.py stack_frame.
"""
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def make_decoders():
    sys.path.insert(0, REF)
    from models.ctc.decoders.greedy_decoder import GreedyDecoder
    from models.ctc.decoders.beam_search_decoder import BeamSearchDecoder
    warnings.simplefilter('ignore')
    rng = np.random.RandomState(1234)
    cases = []
    # (T, C, beam widths, sharpness): peaky posteriors resemble trained CTC output
    specs = [(12, 5, (1, 2, 4), 1.0), (30, 8, (1, 3, 8), 2.0), (50, 29, (1, 5, 20), 3.0),
             (40, 40, (4, 20), 4.0), (25, 62, (10,), 3.0), (16, 6, (20,), 0.3),
             (1, 4, (1, 3), 1.0), (60, 29, (20,), 6.0)]
    out = {}
    n = 0
    for (T, C, widths, sharp) in specs:
        for rep in range(3):
            logits = rng.randn(1, T, C) * sharp
            # make blank (C-1) frequent like a trained model, and add repeats
            logits[0, :, C - 1] += sharp * rng.rand()
            for t in range(1, T):
                if rng.rand() < 0.35:
                    logits[0, t] = logits[0, t - 1] + 0.1 * rng.randn(C)
            probs = _softmax(logits).astype(np.float64)
            seq_len = np.array([T if rep != 2 else max(1, T - 3)], dtype=np.int32)
            g = GreedyDecoder(blank_index=C - 1)(probs, seq_len)
            out['c%d_probs' % n] = probs
            out['c%d_seq_len' % n] = seq_len
            out['c%d_greedy' % n] = np.asarray(g[0], dtype=np.int64)
            for w in widths:
                hyp, score = BeamSearchDecoder(space_index=-1, blank_index=C - 1)(
                    probs, seq_len, beam_width=w)
                out['c%d_beam%d' % (n, w)] = np.asarray(hyp[0], dtype=np.int64)
                out['c%d_beam%d_score' % (n, w)] = np.asarray(score[0], dtype=np.float64)
            out['c%d_widths' % n] = np.array(widths, dtype=np.int64)
            n += 1
    out['num_cases'] = np.array(n)
    np.savez_compressed(os.path.join(HERE, 'decoders_v1.npz'), **out)
    print('decoders_v1.npz: %d cases' % n)


def make_splice():
    sys.path.insert(0, REF)
    # utils/io/inputs/frame_stacking.py imports utils.progressbar -> tqdm; fine here
    from utils.io.inputs.splicing import do_splice
    from utils.io.inputs.frame_stacking import stack_frame
    rng = np.random.RandomState(4321)
    out = {}
    n = 0
    for (T, ch, splice, num_stack) in [(7, 2, 3, 1), (20, 4, 5, 1), (13, 3, 11, 1), (9, 2, 3, 2),
                                       (3, 2, 5, 1), (30, 40, 11, 1)]:
        x = rng.randn(2, T, ch * 3 * num_stack)
        y = do_splice(x, splice=splice, batch_size=2, num_stack=num_stack)
        out['s%d_in' % n] = x
        out['s%d_out' % n] = y
        out['s%d_cfg' % n] = np.array([splice, num_stack])
        n += 1
    out['num_splice'] = np.array(n)
    m = 0
    for (T, D, num_stack, num_skip) in [(10, 6, 2, 2), (11, 6, 3, 3), (7, 3, 3, 2), (5, 4, 2, 1),
                                        (1, 3, 2, 2), (23, 120, 2, 2)]:
        x = rng.randn(T, D)
        y = stack_frame(np.array([x]), num_stack, num_skip)
        out['f%d_in' % m] = x
        out['f%d_out' % m] = np.asarray(y[0])
        out['f%d_cfg' % m] = np.array([num_stack, num_skip])
        m += 1
    out['num_stack'] = np.array(m)
    np.savez_compressed(os.path.join(HERE, 'splice_v1.npz'), **out)
    print('splice_v1.npz: %d splice + %d stack cases' % (n, m))


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs %s (build container only)' % REF)
    make_decoders()
    make_splice()
