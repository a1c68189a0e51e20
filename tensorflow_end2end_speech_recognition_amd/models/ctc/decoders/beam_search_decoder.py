"""Beam search (prefix search) decoder -- mirror of
models/ctc/decoders/beam_search_decoder.py:42-152 (class BeamSearchDecoder(space_index,
blank_index), __call__(probs [B,T,C], seq_len, beam_width=1, alpha=0., beta=0.) ->
(results, scores)), executed by asr_ctc_beam_decode on the GPU.
alpha / beta (LM weight, insertion bonus) are accepted and unused, as in the reference."""
import numpy as np
import torch

from .... import ops
from .greedy_decoder import _to_logits_tbc


class BeamSearchDecoder(object):

    def __init__(self, space_index, blank_index, device='cuda:0'):
        self._space = space_index
        self._blank = blank_index
        self.device = torch.device(device)

    def __call__(self, probs, seq_len, beam_width=1, alpha=0., beta=0.):
        logits = _to_logits_tbc(probs, self.device)
        sl = torch.as_tensor(np.asarray(seq_len), dtype=torch.int32, device=self.device)
        lab, n, score = ops.ctc_beam_decode(logits, sl, int(beam_width), blank=self._blank)
        lab, n = lab.cpu().numpy(), n.cpu().numpy()
        return [lab[b, :n[b]].tolist() for b in range(lab.shape[0])], score.cpu().numpy()
