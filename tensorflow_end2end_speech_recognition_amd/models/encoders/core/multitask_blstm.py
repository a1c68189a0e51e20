"""Multi-task bidirectional LSTM encoder -- mirror of models/encoders/core/multitask_blstm.py:14-125
(class MultitaskBLSTMEncoder): the BLSTM stack of blstm.py:258-332 whose layer `num_layers_sub` output is
returned a second time for the sub-task head (blstm.py:326-328).  Same kernels as BLSTMEncoder; the only new
data flow is the backward join of the two output gradients at that layer.

__call__(inputs, inputs_seq_len, keep_prob, is_training) ->
    (outputs, final_state, outputs_sub, final_state_sub)."""
from .blstm import BLSTMEncoder


class _MultitaskMixin(object):

    def _init_multitask(self, num_layers_main, num_layers_sub):
        if num_layers_sub < 1 or num_layers_main < num_layers_sub:          # multitask_blstm.py:66-68
            raise ValueError('Set num_layers_sub between 1 to num_layers_main.')
        self.num_layers_main = num_layers_main
        self.num_layers_sub = self._effective_sub(num_layers_main, num_layers_sub)
        self.num_layers_sub_arg = num_layers_sub

    def _effective_sub(self, num_layers_main, num_layers_sub):
        return num_layers_sub

    def _call_multitask(self, base_call, inputs, inputs_seq_len, keep_prob, is_training, **kw):
        from .... import ops
        import torch
        outputs, final_state = base_call(inputs, inputs_seq_len, keep_prob, is_training, **kw)
        B = self.batch
        sub = self._out_sub_op
        self._out_sub_tm = ops.cast_to_f32(sub) if sub.dtype != torch.float32 else sub
        final_sub = self._state_tuple_projected(self._finals, self.num_layers_sub) if self.num_proj is not None \
            else self._state_tuple(self._finals, self.num_layers_sub)
        out_sub = self._out_sub_tm[:, :B]
        if not self.time_major:
            out_sub = out_sub.transpose(0, 1)
        return outputs, final_state, out_sub, final_sub


class MultitaskBLSTMEncoder(_MultitaskMixin, BLSTMEncoder):
    """models/encoders/core/multitask_blstm.py:14 MultitaskBLSTMEncoder (same constructor arguments)."""

    def __init__(self, num_units, num_proj, num_layers_main, num_layers_sub, lstm_impl, use_peephole,
                 parameter_init, clip_activation, time_major=False, name='multitask_blstm_encoder', **kw):
        BLSTMEncoder.__init__(self, num_units=num_units, num_proj=num_proj, num_layers=num_layers_main,
                              lstm_impl=lstm_impl, use_peephole=use_peephole, parameter_init=parameter_init,
                              clip_activation=clip_activation, time_major=time_major, name=name, **kw)
        self._init_multitask(num_layers_main, num_layers_sub)

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training=True, **kw):
        return self._call_multitask(lambda *a, **k: BLSTMEncoder.__call__(self, *a, **k), inputs, inputs_seq_len,
                                    keep_prob, is_training, **kw)
