"""CPU, world_size 2 (gloo): the data-parallel exchange of utils/training/multi_gpu.py --
clip-then-mean over ranks == the reference's tower loop (train_ctc.py:112-117,143) -- and the
np.array_split shard rule of utils/dataset/ctc.py:171-182."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import optim as oopt


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tensorflow_end2end_speech_recognition_amd.utils.parameter import ParamStore
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu
    rng = np.random.RandomState(0)
    store = ParamStore(torch.device('cpu'))
    shapes = {'blstm_hidden1/fw/lstm_cell/kernel': (7, 12), 'blstm_hidden1/fw/lstm_cell/bias': (12,),
              'output/weights': (6, 5), 'output/biases': (5,)}
    for n, s in shapes.items():
        store.declare(n, s, rng.randn(*s))
    store.finalize()
    if rank != 0:
        store.flat.add_(1.0)                       # replicas differ before the broadcast
    multi_gpu.broadcast_parameters(store)
    # rank-specific "tower" gradients, clipped locally (train_ctc.py:116), then averaged (:143)
    grng = np.random.RandomState(100 + rank)
    towers = []
    for n in store.names:
        g = grng.randn(*shapes[n]) * 3
        store.g(n).copy_(torch.tensor(oopt.clip_by_norm(g, 1.0), dtype=torch.float32))
    multi_gpu.average_gradients(store)
    loss = multi_gpu.average_scalar(torch.tensor(float(rank + 1)))
    q.put((rank, store.flat.clone().numpy(), store.grad.clone().numpy(), float(loss),
           {n: store.g(n).numpy().copy() for n in store.names}))
    dist.destroy_process_group()


def test_average_gradients_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, p0, g0, l0, gv0), (r1, p1, g1, l1, gv1) = res
    assert np.array_equal(p0, p1)                               # identical replicas after broadcast
    assert np.array_equal(g0, g1)                               # identical averaged gradients
    assert abs(l0 - 1.5) < 1e-6 and abs(l1 - 1.5) < 1e-6
    shapes = {n: v.shape for n, v in gv0.items()}
    towers = []
    for rank in range(world):
        grng = np.random.RandomState(100 + rank)
        towers.append([oopt.clip_by_norm(grng.randn(*shapes[n]) * 3, 1.0) for n in gv0])
    ref = oopt.average_gradients(towers)
    for (n, g), r in zip(gv0.items(), ref):
        assert np.abs(g - r).max() < 1e-6, n


def test_split_batch_rule():
    from tensorflow_end2end_speech_recognition_amd.utils.training.multi_gpu import average_gradients, split_batch
    x = np.arange(10 * 3).reshape(10, 3)
    sl = np.arange(10)
    xs, ss = split_batch([x, sl], 4)
    assert [len(a) for a in xs] == [3, 3, 2, 2] and np.array_equal(np.concatenate(ss), sl)   # np.array_split
    assert np.array_equal(xs[1], x[3:6])
    # reference calling convention: list of per-tower gradient lists, None entries skipped
    t0 = [torch.ones(2, 2), None, torch.full((3,), 2.0)]
    t1 = [torch.full((2, 2), 3.0), None, torch.full((3,), 4.0)]
    avg = average_gradients([t0, t1])
    assert torch.equal(avg[0], torch.full((2, 2), 2.0)) and avg[1] is None and torch.equal(avg[2], torch.full((3,), 3.0))
