import sys, os, ctypes
os.environ['ASR_LSTM_DBG'] = '1'
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import make_batch
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
from tensorflow_end2end_speech_recognition_amd import _lib
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
x, sl, labels, dense = make_batch(1, 16, 120, 62, 100, 778)
m = CTC('blstm', 120, H, 1, 61, dtype='bf16', seed=0)
xd = torch.tensor(x, device=dev); sld = torch.tensor(sl, device=dev)
for it in range(2):
    loss, _ = m.compute_loss(xd, dense, sld, keep_prob=1.0, is_training=True)
    m.train(loss, 'sgd', 0.0)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 256)()
lib.asr_debug_lstm_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
print('rc', lib.asr_debug_lstm_cycles(buf, 256))
a = np.array(list(buf), dtype=np.float64).reshape(2, 16, 8)
names = ['top', 'lds_mfma', 'stream_mfma', 'gate', 'barrier']
for d in range(2):
    steps = a[d, 0, 5]
    print('dir', d, 'steps', steps)
    for w in (0, 1, 4, 8, 15):
        print('  wave %2d: ' % w + '  '.join('%s %.0f' % (n, a[d, w, k] / max(steps, 1)) for k, n in enumerate(names)),
              ' total/step %.0f' % (a[d, w, :5].sum() / max(steps, 1)))

if H in (256, 512):
    buf2 = (ctypes.c_ulonglong * 1280)()
    lib.asr_debug_cluster_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if lib.asr_debug_cluster_cycles(buf2, 1280) == 0:
        allv = np.array(list(buf2), dtype=np.float64)
        for label, a in (('fwd', allv[256:768].reshape(2, 4, 8, 8)), ('bwd', allv[768:1280].reshape(2, 4, 8, 8))):
            names = ['p0', 'p1', 'p2', 'p3']
            for d in range(2):
                for g in (0, 1):
                    for w in range(8):
                        steps = a[d, g, w, 5]
                        if steps == 0:
                            continue
                        print('cluster8', label, 'dir', d, 'cu', g, 'wave', w,
                              ' '.join('%s %.0f' % (n, a[d, g, w, k] / steps) for k, n in enumerate(names)),
                              'total/step %.0f' % (a[d, g, w, :4].sum() / steps), 'fast', a[d, g, w, 6], 'x7/step %.0f' % (a[d, g, w, 7] / steps),
                              'repolls/step %.2f' % (a[d, g, w, 4] / steps))
