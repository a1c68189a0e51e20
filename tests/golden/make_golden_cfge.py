#!/usr/bin/env python
"""Golden vectors of the REFERENCE'S OWN BeamSearchDecoder at BASELINE cfg E scale (C = 3387, beam 20 / 100):
    python tests/golden/make_golden_cfge.py          (build container only: needs /root/reference; ~25 min, one
                                                      process per case -- the reference decoder is pure Python,
                                                      O(T * C * beam) dict operations)
Writes decoders_cfge_v1.json: per case the best prefix, its -log score and the SHA-256 of the fp32 log-posteriors
the device test regenerates from cfge_inputs.py."""
import json
import multiprocessing
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.path.insert(0, HERE)


def run(name):
    sys.path.insert(0, REF)
    warnings.simplefilter('ignore')
    import cfge_inputs
    from models.ctc.decoders.beam_search_decoder import BeamSearchDecoder
    from models.ctc.decoders.greedy_decoder import GreedyDecoder
    probs, sl, W = cfge_inputs.posteriors(name)
    C = probs.shape[2]
    t0 = time.time()
    hyp, score = BeamSearchDecoder(space_index=-1, blank_index=C - 1)(probs, sl, beam_width=W)
    g = GreedyDecoder(blank_index=C - 1)(probs, sl)
    return name, dict(beam_width=W, T=int(sl[0]), C=C, labels=[int(v) for v in hyp[0]], score=float(score[0]),
                      greedy=[int(v) for v in g[0]], logits_sha256=cfge_inputs.digest(cfge_inputs.fp32_logits(probs)),
                      reference_seconds=round(time.time() - t0, 1))


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs %s (build container only)' % REF)
    import cfge_inputs
    names = sys.argv[1:] or list(cfge_inputs.CASES)
    with multiprocessing.Pool(min(len(names), 5)) as pool:
        out = dict(pool.map(run, names))
    path = os.path.join(HERE, 'decoders_cfge_v1.json')
    if os.path.exists(path) and sys.argv[1:]:
        old = json.load(open(path)); old.update(out); out = old
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
    print({k: (len(v['labels']), v['reference_seconds']) for k, v in out.items()})
