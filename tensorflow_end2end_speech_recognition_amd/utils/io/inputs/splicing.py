"""Context splicing for CNN-like encoders -- mirror of utils/io/inputs/splicing.py:9-73 (do_splice),
vectorised; pinned to the reference's outputs by tests/golden/splice_v1.npz.

Reference behaviour kept (quirk Q9: follow the code, not its docstring): frame t is spliced with
frames t-splice .. t-1 (NOT a centred window), the first / last frame replicated at the edges by the
rules of :42-57; each frame [C*3*num_stack] is read as (C, 3, num_stack) and laid out as
[num_channels][splice*num_stack][3] (last fastest), which is what the VGG encoder reshapes
(models/encoders/core/vgg_blstm.py:108-110)."""
import numpy as np


def do_splice(inputs, splice=1, batch_size=1, num_stack=1):
    assert isinstance(inputs, np.ndarray), 'inputs should be np.ndarray.'
    assert len(inputs.shape) == 3, 'inputs must be 3 demension.'
    assert inputs.shape[-1] % 3 == 0
    if splice == 1:
        return inputs
    B, T, D = inputs.shape
    C = (D // 3) // num_stack
    out = np.zeros((B, T, splice * num_stack, C, 3))
    t = np.arange(T)
    for i in range(splice):
        src = t + (i - splice)
        left = (t <= splice - 1) & (i < splice - t)                       # :42-45 copy the first frame
        right = (~left) & (T - splice <= t) & (src > T - 1)               # :50-52 copy the last frame
        src = np.where(left, 0, np.where(right, T - 1, src))
        frames = inputs[:, src].reshape(B, T, C, 3, num_stack)            # :60
        frames = np.transpose(frames, (0, 1, 4, 2, 3))                    # -> [num_stack, C, 3] (:63)
        # the reference assigns spliced_frames[i : i+num_stack] (:65), so later i overwrite the overlap
        hi = min(i + num_stack, splice * num_stack)
        out[:, :, i:hi] = frames[:, :, :hi - i]
    out = np.transpose(out, (0, 1, 3, 2, 4))                              # -> [C, splice*num_stack, 3] (:68)
    return out.reshape(B, T, C * splice * num_stack * 3)
