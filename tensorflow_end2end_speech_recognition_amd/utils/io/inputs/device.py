"""Batch assembly on the device: the frame stacking and context splicing that
utils/dataset/ctc.py:125-166 runs per utterance in Python loops (utils/io/inputs/frame_stacking.py:14-85,
splicing.py:9-73), done by two gather kernels on the zero-padded batch after ONE upload of the raw features
(splice 11 makes the spliced batch 11x the raw one -- it never crosses PCIe this way).

A dataset built with `device_assembly=True` (utils/dataset/ctc.py) yields the raw padded features and raw
frame counts; the recipe then calls assemble() and feeds the result to compute_loss()."""
import numpy as np
import torch

from .... import ops


def assemble(inputs, inputs_seq_len, num_stack=None, num_skip=None, splice=1, device='cuda:0'):
    """inputs [B,Tmax,F] (numpy or tensor, zero-padded), inputs_seq_len [B] ->
    (inputs [B,Tn,F*num_stack*splice] fp32 on `device`, seq_len [B] int32 on `device`): exactly what the host
    path of DatasetBase.__next__ produces for the same utterances."""
    dev = torch.device(device)
    if isinstance(inputs, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).pin_memory().to(dev, non_blocking=True)
    else:
        x = inputs.to(device=dev, dtype=torch.float32).contiguous()
    sl = torch.as_tensor(np.asarray(inputs_seq_len) if not torch.is_tensor(inputs_seq_len) else inputs_seq_len,
                         dtype=torch.int32, device=dev).contiguous()
    stack = 1
    if num_stack is not None and num_skip is not None and num_stack != 1:      # quirk Q10: identity at 1
        x, sl = ops.stack_frames(x, sl, num_stack, num_skip)
        stack = num_stack
    if splice != 1:
        x = ops.splice(x, sl, splice, stack)
    return x, sl
