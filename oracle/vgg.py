"""Oracle (test infrastructure; layout / padding conventions PINNED to TensorFlow's conv_ops / pooling_ops test constants,
the front-end as a whole to the reference's own VGGBLSTMEncoder + cnn_util.py as executed,
tests/test_oracle_tfshim.py::test_tfshim_ctc_models[ctc_vgg_blstm]): CPU restatement of the VGG
front-end of models/encoders/core/vgg_blstm.py:107-177 (conv_layer / max_pool of
models/encoders/core/cnn_util.py:13-84) with TF 'SAME' semantics:
conv 3x3 stride 1 pads 1/1; max_pool 2x2 stride 2 pads the EXTRA cell AFTER with -inf.
Input per frame: [F, W, 3] NHWC (layout of utils/io/inputs/splicing.py:60-73)."""
import torch
import torch.nn.functional as Fn

CONVS = [('VGG1/conv1', 3, 64), ('VGG1/conv2', 64, 64), ('VGG2/conv1', 64, 128), ('VGG2/conv2', 128, 128)]


def _conv(x_nchw, w_hwio, b):
    return Fn.relu(Fn.conv2d(x_nchw, w_hwio.permute(3, 2, 0, 1), b, padding=1))


def _pool_same(x):
    H, W = x.shape[2], x.shape[3]
    ph, pw = H % 2, W % 2                      # total pad for k=2,s=2: (ceil(n/2)-1)*2+2-n
    x = Fn.pad(x, (0, pw, 0, ph), value=float('-inf'))
    return Fn.max_pool2d(x, 2, 2)


def vgg_frontend(x_btd, P, F, W, act_round=None):
    """x [B,T,F*W*3] -> [B,T,256].  P: dict of torch tensors with the reference variable names.
    act_round (tests of the bf16-operand device path): rounding applied, straight-through, to every activation
    the device stores in the operand dtype (the four ReLU outputs and the bridge output; pooling is exact on
    rounded values).  The caller rounds the inputs and the weights."""
    from .lstm import ste_round
    r = (lambda t: t) if act_round is None else (lambda t: ste_round(t, act_round))
    B, T, _ = x_btd.shape
    x = x_btd.reshape(B * T, F, W, 3).permute(0, 3, 1, 2)
    x = r(_conv(x, P['VGG1/conv1/weight'], P['VGG1/conv1/bias']))
    x = r(_conv(x, P['VGG1/conv2/weight'], P['VGG1/conv2/bias']))
    x = _pool_same(x)
    x = r(_conv(x, P['VGG2/conv1/weight'], P['VGG2/conv1/bias']))
    x = r(_conv(x, P['VGG2/conv2/weight'], P['VGG2/conv2/bias']))
    x = _pool_same(x)
    x = x.permute(0, 2, 3, 1).reshape(B * T, -1)          # NHWC flatten (vgg_blstm.py:160-161)
    x = r(Fn.relu(x @ P['bridge/weights'] + P['bridge/biases']))
    return x.reshape(B, T, -1)
