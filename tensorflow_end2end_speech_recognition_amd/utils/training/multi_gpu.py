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
    if is_distributed():
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
        if device.type == 'cuda':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
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
