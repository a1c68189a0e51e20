"""Bridges between the encoder's final state and the decoder's initial state -- mirror of
models/attention/bridge.py:28-151 (Bridge, ZeroBridge, PassThroughBridge, InitialStateBridge).

InitialStateBridge (the one AttentionSeq2Seq uses, attention_seq2seq.py:229-236): the final (c, h) of the last
encoder layer, forward then backward direction, flattened and concatenated on the depth axis -> one fully connected
layer (identity activation, truncated-normal weights, zero bias; variables bridge/fully_connected/{weights,biases})
-> split into the decoder's (c0, h0).  encoder_outputs: an object with `.final_state` = ((c_fw, h_fw), (c_bw, h_bw))
as the encoders here return it (LSTMStateTuple pairs), and `.outputs` for the batch size."""
import numpy as np
import torch

from ... import ops as _ops


def _flatten_state(fs):
    out = []
    for t in fs:
        if isinstance(t, (tuple, list)):
            out.extend(_flatten_state(t))
        else:
            out.append(t)
    return out


class Bridge(object):
    def __init__(self, encoder_outputs, decoder_state_size):
        self.encoder_outputs = encoder_outputs
        self.decoder_state_size = decoder_state_size
        self.batch_size = _flatten_state([encoder_outputs.final_state])[0].shape[0]

    def __call__(self):
        return self._create()

    def _create(self):
        raise NotImplementedError


class ZeroBridge(Bridge):
    """bridge.py:64-78: zero initial state."""

    def _create(self):
        dev = _flatten_state([self.encoder_outputs.final_state])[0].device
        return tuple(torch.zeros((self.batch_size, s), dtype=torch.float32, device=dev)
                     for s in _sizes(self.decoder_state_size))


class PassThroughBridge(Bridge):
    """bridge.py:81-94: the encoder's final state IS the decoder's initial state (shapes must agree)."""

    def _create(self):
        flat = _flatten_state([self.encoder_outputs.final_state])
        want = _sizes(self.decoder_state_size)
        if [t.shape[1] for t in flat] != list(want):
            raise ValueError('PassThroughBridge: encoder final state %s does not match the decoder state sizes %s'
                             % ([tuple(t.shape) for t in flat], list(want)))
        return tuple(flat)


def _sizes(state_size):
    return [int(s) for s in (state_size if isinstance(state_size, (tuple, list)) else [state_size])]


class InitialStateBridge(Bridge):
    """bridge.py:97-151."""

    def __init__(self, encoder_outputs, decoder_state_size, parameter_init, store=None, seed=0):
        super(InitialStateBridge, self).__init__(encoder_outputs, decoder_state_size)
        if not hasattr(encoder_outputs, 'final_state'):
            raise ValueError('Invalid bridge_input not in encoder outputs.')
        self._bridge_input = encoder_outputs.final_state
        self.parameter_init = parameter_init
        self.store = store
        self.seed = seed

    def _ensure_vars(self, din, dout, device):
        if self.store is not None:
            return
        from ...utils.parameter import ParamStore
        from ..ctc.ctc import truncated_normal
        rng = np.random.RandomState(self.seed)
        st = self.store = ParamStore(device)
        st.declare('bridge/fully_connected/weights', (din, dout), truncated_normal(rng, self.parameter_init, (din, dout)))
        st.declare('bridge/fully_connected/biases', (dout,), np.zeros(dout))
        st.finalize()

    def bridge_input(self):
        """[B, total depth]: every tensor of the final state reshaped to [B, depth] and concatenated (:130-135)."""
        flat = [t.reshape(self.batch_size, -1) for t in _flatten_state([self._bridge_input])]
        return torch.cat(flat, dim=1).contiguous()

    def _create(self):
        bi = self.bridge_input()
        splits = _sizes(self.decoder_state_size)
        self._ensure_vars(bi.shape[1], sum(splits), bi.device)
        st = self.store
        init = _ops.gemm(bi, st['bridge/fully_connected/weights'], bias=st['bridge/fully_connected/biases'])
        out, o = [], 0
        for s in splits:
            out.append(init[:, o:o + s].contiguous())
            o += s
        return tuple(out)
