"""Device batch assembly (asr_stack_frames + asr_splice) on a cfg-C-shaped batch: achieved HBM rate of the splice
kernel (algorithmic bytes = raw read once + spliced write once) next to the host path of DatasetBase (numpy
do_splice per utterance), timed on a bounded sample of the same utterances.  Usage: python scripts/probe_assembly.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_end2end_speech_recognition_amd import ops                                   # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.device import assemble      # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.splicing import do_splice   # noqa: E402


def main():
    rng = np.random.RandomState(2)
    B, D, splice = 64, 120, 11
    lens = rng.randint(150, 1651, size=B).astype(np.int32)
    T = int(lens.max())
    raw = np.zeros((B, T, D), dtype=np.float32)
    for b in range(B):
        raw[b, :lens[b]] = rng.randn(lens[b], D)
    dev = torch.device('cuda:0')
    x = torch.tensor(raw, device=dev)
    sl = torch.tensor(lens, device=dev)
    out = ops.splice(x, sl, splice, 1)
    torch.cuda.synchronize()
    # correctness on the sample the host path is timed on
    sample = [0, 1, 2, 3]
    t0 = time.perf_counter()
    host = [do_splice(raw[b:b + 1, :lens[b]].astype(np.float64), splice, 1, 1)[0].astype(np.float32) for b in sample]
    host_s = time.perf_counter() - t0
    got = out.cpu().numpy()
    ok = all(np.array_equal(got[b, :lens[b]], host[i]) and not got[b, lens[b]:].any() for i, b in enumerate(sample))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.splice(x, sl, splice, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alg_bytes = raw.nbytes + raw.nbytes * splice
    # upload of the raw batch + assembly on the device, end to end from a host array
    t0 = time.perf_counter()
    for _ in range(5):
        xa, sa = assemble(raw, lens, None, None, splice, device=dev)
    torch.cuda.synchronize()
    e2e_ms = (time.perf_counter() - t0) / 5 * 1e3
    host_frames = int(sum(lens[b] for b in sample))
    print(json.dumps(dict(batch=[B, T, D], splice=splice, bit_exact_on_sample=bool(ok), splice_kernel_ms=ms,
                          algorithmic_GB=alg_bytes / 1e9, achieved_GBps=alg_bytes / 1e9 / (ms * 1e-3),
                          frac_of_8TBps=alg_bytes / 1e9 / (ms * 1e-3) / 8000.0,
                          upload_plus_assemble_ms=e2e_ms, frames=int(lens.sum()),
                          host_numpy_frames_per_s=host_frames / host_s,
                          device_frames_per_s_kernel_only=float(lens.sum()) / (ms * 1e-3))))


if __name__ == '__main__':
    main()
