"""Select & load encoder -- mirror of models/encoders/load_encoder.py:26-57.

The registry is the reference's plugin point: an encoder is a class with
__call__(inputs, inputs_seq_len, keep_prob, is_training) -> (outputs, final_state).
Keys outside the hot path and its "next" rows (cnn_zhang, vgg_wang, pyramid_blstm, student_*) are not built
(SURVEY.md section 2 row 6): asking for them raises the same ValueError as an unknown key."""
from .core.blstm import BLSTMEncoder
from .core.lstm import LSTMEncoder
from .core.vgg_blstm import VGGBLSTMEncoder, VGGLSTMEncoder
from .core.multitask_blstm import MultitaskBLSTMEncoder
from .core.multitask_lstm import MultitaskLSTMEncoder
from .core.gru import GRUEncoder, BGRUEncoder
from .core.cldnn_wang import CLDNNEncoder

ENCODERS = {
    "blstm": BLSTMEncoder,
    "lstm": LSTMEncoder,
    "vgg_blstm": VGGBLSTMEncoder,
    "vgg_lstm": VGGLSTMEncoder,
    "multitask_blstm": MultitaskBLSTMEncoder,     # the same kernels recombined
    "multitask_lstm": MultitaskLSTMEncoder,
    "bgru": BGRUEncoder,
    "gru": GRUEncoder,
    "cldnn_wang": CLDNNEncoder,
}


def register(encoder_type, cls):
    ENCODERS[encoder_type] = cls


def load(encoder_type):
    if encoder_type not in ENCODERS.keys():
        raise ValueError(
            "encoder_type should be one of [%s], you provided %s." %
            (", ".join(ENCODERS), encoder_type))
    return ENCODERS[encoder_type]
