"""Multi-task unidirectional LSTM encoder -- mirror of models/encoders/core/multitask_lstm.py:14-125 over the
stack of lstm.py:241-304.

Reference behaviour kept as coded: `lstm_list_sub = lstm_list` (lstm.py:271-272) aliases the list that keeps
growing, so the "sub" MultiRNNCell built at :290-291 holds ALL num_layers_main cells and -- under reuse=True --
the sub-task outputs are a second evaluation of the FULL stack with the same weights: for keep_prob = 1 they equal
the main outputs whatever num_layers_sub says.  Here that second evaluation is not repeated: the sub outputs are
the main outputs (and their gradients add at the top layer).  With dropout the reference's second pass would draw
its own masks; this build shares the pass (flagged deviation)."""
from .lstm import LSTMEncoder
from .multitask_blstm import _MultitaskMixin


class MultitaskLSTMEncoder(_MultitaskMixin, LSTMEncoder):
    """models/encoders/core/multitask_lstm.py:14 MultitaskLSTMEncoder (same constructor arguments)."""

    def __init__(self, num_units, num_proj, num_layers_main, num_layers_sub, lstm_impl, use_peephole,
                 parameter_init, clip_activation, time_major=False, name='multitask_lstm_encoder', **kw):
        LSTMEncoder.__init__(self, num_units=num_units, num_proj=num_proj, num_layers=num_layers_main,
                             lstm_impl=lstm_impl, use_peephole=use_peephole, parameter_init=parameter_init,
                             clip_activation=clip_activation, time_major=time_major, name=name, **kw)
        self._init_multitask(num_layers_main, num_layers_sub)

    def _effective_sub(self, num_layers_main, num_layers_sub):
        return num_layers_main                    # the list alias of lstm.py:271-272

    def __call__(self, inputs, inputs_seq_len, keep_prob, is_training=True, **kw):
        return self._call_multitask(lambda *a, **k: LSTMEncoder.__call__(self, *a, **k), inputs, inputs_seq_len,
                                    keep_prob, is_training, **kw)
