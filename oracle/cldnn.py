"""Oracle (test infrastructure, PARITY UNPINNED -- TF1 absent; the convolution / padding conventions are the ones
oracle/vgg.py pins to TensorFlow's conv_ops_test known answers): CPU restatement of the CLDNN encoder of
models/encoders/core/cldnn_wang.py:134-249 -- three SAME convolutions (11x21 stride (3,2), 11x11 stride (1,2), 3x3) with
ReLU (the 1x1 max_pools are identities), the BLSTM stack, fc1 896 relu, fc2 74 relu."""
import torch
import torch.nn.functional as Fn

CONVS = [('CNN1/conv', (3, 2)), ('CNN2/conv', (1, 2)), ('CNN3/conv', (1, 1))]


def conv_same(x_nchw, w_hwio, b, stride):
    """tf.nn.conv2d(padding='SAME'): out = ceil(in / stride); the odd padding cell goes AFTER."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    H, W = x_nchw.shape[2], x_nchw.shape[3]
    sh, sw = stride
    ph = max((-(-H // sh) - 1) * sh + kh - H, 0)
    pw = max((-(-W // sw) - 1) * sw + kw - W, 0)
    x = Fn.pad(x_nchw, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return Fn.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, stride=stride)


def conv_stack(x_btd, P, F, W, masks=None, act_round=None):
    """x [B,T,F*W*3] -> [B,T,h*w*96].  masks: per-block dropout masks [B*T,h,w,C] (NHWC) or None.
    act_round: rounding (straight-through) of every stored activation, as the bf16-operand device path does."""
    from .lstm import ste_round
    B, T, _ = x_btd.shape
    x = x_btd.reshape(B * T, F, W, 3).permute(0, 3, 1, 2)
    for i, (name, stride) in enumerate(CONVS):
        x = torch.relu(conv_same(x, P[name + '/weight'], P[name + '/bias'], stride))
        if act_round is not None:
            x = ste_round(x, act_round)
        if masks is not None and masks[i] is not None:
            x = x * masks[i].permute(0, 3, 1, 2)
    return x.permute(0, 2, 3, 1).reshape(B, T, -1)
