"""A few training steps of the bench's blstmp entry (blstm 5x256, LSTMCell, num_proj 128, bf16 operands, headline batch) for a
kernel trace: rocprofv3 --kernel-trace --stats -- python scripts/probe_lstmp_step.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC  # noqa: E402

dev = torch.device('cuda:0')
x, sl, _, dense = bench.make_batch(1, 16, 120, 62, 100, 778)
xd, sd = torch.tensor(x, device=dev), torch.tensor(sl, device=dev)
m = CTC('blstm', 120, 256, 5, 61, lstm_impl='LSTMCell', num_proj=128, clip_grad_norm=5.0, clip_activation=50.0, seed=0,
        dtype=os.environ.get('DTYPE', 'bf16'), device=str(dev))
for it in range(int(os.environ.get('STEPS', '8'))):
    loss, _ = m.compute_loss(xd, dense, sd, keep_prob=0.8)
    m.train(loss, 'rmsprop', 1e-3)
torch.cuda.synchronize()
print('loss %.4f' % loss.item())
from tensorflow_end2end_speech_recognition_amd import ops  # noqa: E402
print('cluster hand-off flags', ops.check_async_errors(0))
