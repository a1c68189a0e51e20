#!/usr/bin/env python
"""Greedy attention inference (asr_att_decoder_infer) at the cfg D / cfg E shapes of bench.py: wall time per call and,
under `rocprofv3 --kernel-trace --stats`, the kernels of a call.  WHICH=D|E (default E)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from bench import device_features
from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention

which = os.environ.get('WHICH', 'E')
dev = torch.device('cuda:0')
if which == 'D':
    seed, D, C, att, tlo, thi, ldiv = 3, 240, 28, 'location', 100, 1600, 4
else:
    seed, D, C, att, tlo, thi, ldiv = 4, 246, 3386, 'hybrid', 100, 1000, 6
B, H, L, U, A, Em = 32, 512, 5, 512, 128, 64
rng = np.random.RandomState(seed)
seq_len = rng.randint(tlo, thi + 1, size=B).astype(np.int32)
lens = np.maximum(1, seq_len // ldiv)
Lmax = int(lens.max()) + 2
xd = device_features(seed, seq_len, D, dev)
model = JointCTCAttention(input_size=D, encoder_type='blstm', encoder_num_units=H, encoder_num_layers=L,
                          encoder_num_proj=None, attention_type=att, attention_dim=A, decoder_type='lstm',
                          decoder_num_units=U, decoder_num_layers=1, embedding_dim=Em, lambda_weight=0.5,
                          num_classes=C, sos_index=C, eos_index=C + 1, max_decode_length=Lmax, parameter_init=0.1,
                          clip_grad_norm=5.0, clip_activation_encoder=50, clip_activation_decoder=50, dtype='bf16',
                          seed=5, device=str(dev))
model.infer(xd, seq_len)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    ids = model.infer(xd, seq_len)
    t = time.perf_counter() - t0
    print('cfg %s infer: %.1f ms per call, %d steps issued, ids %s -> %.0f tokens/s' % (
        which, t * 1e3, int(model._infer_raw['steps_issued']), ids.shape, B * ids.shape[1] / t))
