"""Synchronous data-parallel gradient averaging -- the MI355X form of
utils/training/multi_gpu.py:13-48 (average_gradients) + the tower loop of
examples/librispeech/training/train_ctc.py:82-147.

Reference: ONE process, N in-graph towers; per tower compute_gradients and per-variable
clip_by_norm (train_ctc.py:112-117), then per variable stack the N tower gradients and
reduce_mean (multi_gpu.py:32-40), one apply_gradients.
Here: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI); every rank
holds an identical replica, clips locally, then ONE all-reduce(sum) over the flat fp32
gradient buffer of the ParamStore (a single contiguous bucket: 28 MB for the 5x256 BLSTM)
followed by a 1/N scale -- the same mean over towers; every rank then applies the identical
optimizer step, so replicas stay bit-identical without a broadcast after step 0.
"""
import torch
import torch.distributed as dist

from ... import ops


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_parameters(store, src=0):
    """Make replicas identical once, before the first step."""
    if is_distributed():
        dist.broadcast(store.flat, src=src)
        store.mark_dirty()


_native_comm = {}      # device index -> ops.NativeComm, or False after a failed bootstrap


def native_comm(device):
    """The RCCL communicator of the C ABI (asr_comm_init / asr_allreduce_mean) for this rank's GPU, created on first
    use: rank 0 draws the unique id, torch.distributed (already up for the launcher's rendezvous) carries the 128
    bytes to the other ranks.  Returns None when not distributed, on a CPU device, when ASR_DP_COLLECTIVE=torch, or
    when the bootstrap failed (warned once; the torch.distributed all-reduce -- the same RCCL -- is used instead)."""
    import os
    import warnings
    device = torch.device(device)
    if device.type != 'cuda' or not is_distributed() or os.environ.get('ASR_DP_COLLECTIVE', '') == 'torch':
        return None
    key = device.index or 0
    if key not in _native_comm:
        rank, world = dist.get_rank(), dist.get_world_size()
        comm, err = None, None
        try:
            box = [ops.NativeComm.unique_id() if rank == 0 else None]
        except Exception as e:      # keep the collective below matched on every rank
            box, err = [None], e
        dist.broadcast_object_list(box, src=0)
        ok = torch.zeros(1, dtype=torch.int32, device=device)
        if box[0] is not None:
            try:
                comm = ops.NativeComm(key, rank, world, box[0])
                ok += 1
            except Exception as e:
                err = e
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        if int(ok.item()) != world:      # all or nothing: a partial communicator would deadlock the first all-reduce
            if comm is not None:
                comm.close()
            comm = False
            warnings.warn('native RCCL communicator not available (%s); using torch.distributed all_reduce' % (err,))
        _native_comm[key] = comm
    return _native_comm[key] or None


def average_gradients(store_or_tower_grads):
    """ParamStore -> in-place mean of store.grad over all ranks.
    (A list of per-tower gradient lists, the reference's calling convention, is averaged
    on the host side of whatever device the tensors live on.)"""
    if isinstance(store_or_tower_grads, (list, tuple)):
        out = []
        for grads in zip(*store_or_tower_grads):
            gs = [g for g in grads if g is not None]          # multi_gpu.py:30-36 skips None
            out.append(torch.stack(gs, 0).mean(0) if gs else None)
        return out
    store = store_or_tower_grads
    comm = native_comm(store.grad.device) if store.grad.is_cuda else None
    if comm is not None:
        comm.allreduce_mean(store.grad)         # RCCL through the C ABI: all-reduce(sum) + x 1/N on this stream
    elif is_distributed():
        dist.all_reduce(store.grad, op=dist.ReduceOp.SUM)
        n = dist.get_world_size()
        if store.grad.is_cuda:
            ops.scale_(store.grad, 1.0 / n)
        else:
            store.grad.mul_(1.0 / n)
    return store.grad


def average_scalar(x):
    """loss / LER are tower-averaged too (train_ctc.py:136-139)."""
    if is_distributed():
        t = x.detach().clone().float()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / dist.get_world_size()
    return x


def split_batch(arrays, num_gpu):
    """utils/dataset/ctc.py:171-182: np.array_split of every batch tensor along axis 0."""
    import numpy as np
    return [np.array_split(a, num_gpu, axis=0) for a in arrays]


def init_process_group(device):
    """One process per GPU: join the job `python -m torch.distributed.run` started (RANK / WORLD_SIZE / MASTER_* in
    the environment).  RCCL ("nccl") for a GPU device, gloo for a CPU one (tests).  Returns (rank, world)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        device = torch.device(device)
        # rank 0 alone evaluates dev / test sets at epoch ends while the others wait in a collective
        # (examples/librispeech/training/train_ctc.py): far longer than the default 10-minute watchdog
        import datetime
        timeout = datetime.timedelta(hours=float(os.environ.get('ASR_DIST_TIMEOUT_HOURS', '6')))
        if device.type == 'cuda':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=timeout)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timeout)
    return rank, world


def tower_step(model, optimizer, inputs, labels, inputs_seq_len, keep_prob, learning_rate=None):
    """One synchronous data-parallel step of THIS rank's tower -- the body of the tower loop of
    examples/librispeech/training/train_ctc.py:82-147 plus its apply_gradients:
    compute_loss -> compute_gradients -> per-variable clip_by_norm (:116, before the mean) -> mean over towers
    (utils/training/multi_gpu.py:13-48) -> the identical optimizer update on every rank.
    A rank whose shard of the global batch is empty (np.array_split of a short last batch; the reference does not
    guard this) contributes zero gradients and loss 0 -- it must still take part in the collective.
    Returns (loss averaged over towers, this tower's logits or None)."""
    B = len(inputs)
    logits = None
    if B > 0:
        loss, logits = model.compute_loss(inputs, labels, inputs_seq_len, keep_prob)
        gv = optimizer.compute_gradients(loss, model=model)
        if model.clip_grad_norm is not None:
            model._clip_gradients(gv)
        loss = loss.detach()
    else:
        model.store.grad.zero_()
        loss = torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
    average_gradients(model.store)
    optimizer.apply_gradients(None, learning_rate=learning_rate)
    return average_scalar(loss), logits


def tower_step_with(model, optimizer, loss_fn, learning_rate=None):
    """tower_step for any model family: `loss_fn()` runs this rank's forward and returns the loss tensor of
    compute_loss (or None when the rank's shard of the global batch is empty); the rest is the tower loop of
    examples/csj/training/train_attention.py:90-150 -- gradients, per-variable clip on the tower, mean over towers,
    identical update.  Returns the loss averaged over towers."""
    loss = loss_fn()
    if loss is not None:
        gv = optimizer.compute_gradients(loss, model=model)
        if model.clip_grad_norm is not None:
            model._clip_gradients(gv)
        loss = loss.detach()
    else:
        model.store.grad.zero_()
        loss = torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
    average_gradients(model.store)
    optimizer.apply_gradients(None, learning_rate=learning_rate)
    return average_scalar(loss)
