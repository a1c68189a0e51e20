"""16-byte per-lane stores against 16-byte polling loads of another CU: are they ever seen torn?  (asr_debug_tear_probe;
evidence for DESIGN section 8 item 1-i -- the product path does not rely on it.)"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tensorflow_end2end_speech_recognition_amd import _lib  # noqa: E402

h = _lib.handle(0, 0)
dev = torch.device('cuda:0')
buf = torch.zeros(64 * 4, dtype=torch.int32, device=dev)
out = torch.zeros(64 * 3, dtype=torch.int64, device=dev)
iters = int(os.environ.get('ITERS', '20000000'))
for name, peer, wt in (('same XCD, plain stores', 8, 0), ('same XCD, write-through stores', 8, 1),
                       ('other XCD, write-through stores', 1, 1)):
    tot = [0, 0, 0]
    t0 = time.time()
    for rep in range(int(os.environ.get('REPS', '3'))):
        out.zero_()
        rc = h.lib.asr_debug_tear_probe(h.h, C.c_void_p(buf.data_ptr()), C.c_uint(iters), peer, wt,
                                        C.c_void_p(out.data_ptr()), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        o = out.view(64, 3).cpu()
        for k in range(3):
            tot[k] += int(o[:, k].sum())
    print('%-34s %d stores per lane x %d runs: %12d lane-loads, %d torn, %d distinct values seen, %.1f s'
          % (name, iters, int(os.environ.get('REPS', '3')), tot[0], tot[1], tot[2], time.time() - t0), flush=True)
