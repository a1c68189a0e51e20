"""Host-side timeline of the bench step (no extra syncs): when does the host enter/leave each stage,
relative to step start, next to the GPU-synchronised step time.  Finds host-bound stretches."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import make_batch
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
from tensorflow_end2end_speech_recognition_amd import ops
dev = torch.device('cuda:0')
x, sl, labels, dense = make_batch(1, 16, 120, 62, 100, 778)
m = CTC('blstm', 120, 256, 5, 61, clip_grad_norm=5.0, clip_activation=50, dtype='bf16', seed=0)
xd = torch.tensor(x, device=dev); sld = torch.tensor(sl, device=dev)
opt = m._set_optimizer('rmsprop', 1e-3)
marks = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.append((tag + '>', time.perf_counter()))
        r = f(*a, **k)
        marks.append((tag + '<', time.perf_counter()))
        return r
    setattr(obj, name, g)
wrap(m, '_build', 'build'); wrap(m, '_labels_to_flat', 'labels'); wrap(ops, 'ctc_loss', 'ctc')
wrap(m, '_backward', 'backward'); wrap(m, '_clip_gradients', 'clip'); wrap(opt, 'apply_gradients', 'apply')
def step():
    loss, logits = m.compute_loss(xd, dense, sld, keep_prob=0.8)
    gv = opt.compute_gradients(loss, model=m)
    m._clip_gradients(gv)
    opt.apply_gradients(gv)
for _ in range(3): step()
torch.cuda.synchronize()
N = 8
marks.clear()
t0 = time.perf_counter()
starts = []
for _ in range(N):
    starts.append(time.perf_counter()); step()
torch.cuda.synchronize()
t1 = time.perf_counter()
print('wall per step %.2f ms' % ((t1 - t0) / N * 1e3))
per = len(marks) // N
for i in (2, N - 1):
    print('step', i, 'host start at %.2f ms' % ((starts[i] - t0) * 1e3))
    for tag, t in marks[i * per:(i + 1) * per]:
        print('   %-10s +%.2f ms' % (tag, (t - starts[i]) * 1e3))
    end = starts[i + 1] if i + 1 < N else t1
    print('   step end   +%.2f ms' % ((end - starts[i]) * 1e3))
