"""Greedy (best path) decoder -- mirror of models/ctc/decoders/greedy_decoder.py:13-50
(class GreedyDecoder(blank_index), __call__(probs [B,T,C], seq_len [B]) -> list of label lists),
executed by asr_ctc_greedy_decode on the GPU.  The reference takes np.log(probs) first;
argmax is unchanged by the log, so the kernel runs on log(probs) directly."""
import numpy as np
import torch

from .... import ops


def _to_logits_tbc(probs, device):
    p = torch.as_tensor(np.asarray(probs), dtype=torch.float64)
    return torch.log(p).to(torch.float32).transpose(0, 1).contiguous().to(device)


class GreedyDecoder(object):

    def __init__(self, blank_index, device='cuda:0'):
        self.blank = blank_index
        self.device = torch.device(device)

    def __call__(self, probs, seq_len):
        logits = _to_logits_tbc(probs, self.device)
        sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.int32, device=self.device)
        lab, n = ops.ctc_greedy_decode(logits, sl, blank=self.blank)
        lab, n = lab.cpu().numpy(), n.cpu().numpy()
        # the reference returns np.array(results), which is ragged for B > 1; a list of lists is
        # the usable equivalent (its own B>1 call raises on numpy >= 1.24)
        return [lab[b, :n[b]].tolist() for b in range(lab.shape[0])]
