"""Does the BPTT kernel's launch time depend on where its saved activations live?  One 256-unit bf16 BLSTM layer at the
headline shape: forward, then the backward kernel timed (a) right after the forward (operands in the memory-side cache),
(b) after 1 GB of unrelated traffic (operands in HBM only), (c) again at once (warm)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tensorflow_end2end_speech_recognition_amd import ops
from tensorflow_end2end_speech_recognition_amd.models.encoders.core.rnn_util import LSTMLayer, declare_lstm_vars
from tensorflow_end2end_speech_recognition_amd.utils.parameter import ParamStore
from tensorflow_end2end_speech_recognition_amd._lib import ASR_BF16
dev = torch.device('cuda:0')
T, B, H, D = 778, 16, int(os.environ.get('PH', 256)), 512
rng = np.random.RandomState(0)
st = ParamStore(dev)
bases = declare_lstm_vars(st, 'blstm_hidden1', D, H, 2, True, 0.1, rng)
layer = LSTMLayer(st, bases, D, H, True, 1.0, 50.0)
st.finalize()
sl = torch.tensor(rng.randint(100, T + 1, size=B).astype(np.int32), device=dev); sl[0] = T
x = torch.tensor(rng.randn(T, B, D) * 0.5, dtype=torch.float32, device=dev).to(torch.bfloat16)
dout = torch.tensor(rng.randn(T, B, 2 * H) * 0.1, dtype=torch.float32, device=dev)
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GB


def bwd_time(flush):
    out, _ = layer.forward(x, sl, ASR_BF16, 1.0, False)
    c = layer.ctx
    torch.cuda.synchronize()
    if flush:
        junk.fill_(1.0); junk.mul_(1.0001)
        torch.cuda.synchronize()
    ts = []
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.lstm_bwd(dout, c['gates'], c['cs'], c['whb'], c['peep'], c['seq_len'], H, 2, ASR_BF16, None, None, want_dpeep=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return ts


for it in range(3):
    a = bwd_time(False); b = bwd_time(True)
    print('H=%d  after forward: %.0f us, again %.0f us | after 1 GB of other traffic: %.0f us, again %.0f us' % (H, a[0], a[1], b[0], b[1]))

# ---- the forward kernel: x-projection just written by the GEMM (never read) vs read once before the launch
prep = layer.prepare(dev, ASR_BF16, T, B, 1.0, False, None, None, ldk=D)
for it in range(3):
    ts = []
    for warm in (False, True):
        xproj = torch.empty((T, B, 2 * 4 * H), dtype=torch.float32, device=dev)
        junk.mul_(1.0001)
        ops.gemm(x.view(T * B, D), prep['wxT'], transB=True, bias=prep['bias'], out=xproj.view(T * B, 8 * H))
        if warm:
            ops.touch(xproj)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.lstm_fwd(xproj, prep['whf'], prep['peep'], sl, H, 2, ASR_BF16, 1.0, 50.0)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print('H=%d forward kernel: x-projection fresh from the GEMM %.0f us, after a read pass %.0f us' % (H, ts[0], ts[1]))
