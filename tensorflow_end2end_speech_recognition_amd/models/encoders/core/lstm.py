"""Unidirectional LSTM encoder -- mirror of models/encoders/core/lstm.py:13-116
(MultiRNNCell of LSTMBlockCell under one dynamic_rnn, lstm.py:241-304).
Variable names: multi_lstm/multi_rnn_cell/cell_{i-1}/lstm_cell/... (Appendix C)."""
from .blstm import _RecurrentEncoderBase
from .rnn_util import declare_lstm_vars, declare_lstmp_vars


class LSTMEncoder(_RecurrentEncoderBase):
    ndir = 1

    def _declare(self, store, i, din, rng):
        cell = '%smulti_lstm/multi_rnn_cell/cell_%d/lstm_cell' % (self.scope_prefix, i - 1)
        return declare_lstm_vars(store, None, din, self.num_units, 1, self.use_peephole,
                                 self.parameter_init, rng, cell_scope=cell)

    def _declare_projected(self, store, i, din, P, rng):
        cell = '%smulti_lstm/multi_rnn_cell/cell_%d/lstm_cell' % (self.scope_prefix, i - 1)
        return declare_lstmp_vars(store, None, din, self.num_units, P, 1, self.use_peephole, self.parameter_init, rng,
                                  cell_scope=cell)
