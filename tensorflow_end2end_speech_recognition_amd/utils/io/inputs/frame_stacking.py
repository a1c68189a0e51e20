"""Frame stacking / skipping (Sak et al. 2015) -- mirror of utils/io/inputs/frame_stacking.py:14-85
(stack_frame), restated as index arithmetic instead of the reference's per-frame Python stack;
pinned to the reference's outputs by tests/golden/splice_v1.npz.

Behaviour of the reference loop, kept exactly: output frame k (k = 0 .. ceil(T/num_skip)-1) holds
input frames k*num_skip .. k*num_skip+num_stack-1 side by side; slots that would run past the last
input frame stay zero (the final-frame flush at :45-60 writes only the frames still in the stack).
Quirk Q10 (`num_stack == 1 and num_stack == 1`, :28) -> identity when num_stack == 1."""
import math

import numpy as np


def stack_frame(input_list, num_stack, num_skip, progressbar=False):
    if num_stack == 1:
        return input_list
    if num_stack < num_skip:
        raise ValueError('num_skip must be less than num_stack.')
    out = []
    for x in input_list:
        x = np.asarray(x)
        T, D = x.shape
        Tn = int(math.ceil(T / num_skip))
        y = np.zeros((Tn, D * num_stack), dtype=np.float64)
        for i in range(num_stack):
            src = np.arange(Tn) * num_skip + i
            ok = src < T
            y[ok, D * i:D * (i + 1)] = x[src[ok]]
        out.append(y)
    return np.array(out, dtype=object) if len({o.shape for o in out}) > 1 else np.array(out)
