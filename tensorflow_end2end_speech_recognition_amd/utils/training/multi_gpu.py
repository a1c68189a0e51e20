"""Synchronous data-parallel gradient averaging -- the MI355X form of
utils/training/multi_gpu.py:13-48 (average_gradients) + the tower loop of
examples/librispeech/training/train_ctc.py:82-147.

Reference: ONE process, N in-graph towers; per tower compute_gradients and per-variable
clip_by_norm (train_ctc.py:112-117), then per variable stack the N tower gradients and
reduce_mean (multi_gpu.py:32-40), one apply_gradients.
Here: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI); every rank
holds an identical replica, clips locally, then all-reduces(sum) the flat fp32 gradient buffer
of the ParamStore and scales by 1/N -- the same mean over towers -- per encoder layer on a
communication stream while the layers below are still in their BPTT (BucketedAverager), or as
ONE bucket (average_gradients: 28 MB for the 5x256 BLSTM) where that does not apply; every rank
then applies the identical optimizer step, so replicas stay bit-identical without a broadcast
after step 0.
"""
import os

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


def _allreduce_mean_(t):
    """In-place mean over ranks of a contiguous fp32 tensor (view of the flat gradient buffer), on the current stream."""
    comm = native_comm(t.device) if t.is_cuda else None
    if comm is not None:
        comm.allreduce_mean(t)
    elif is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n = dist.get_world_size()
        if t.is_cuda:
            ops.scale_(t, 1.0 / n)
        else:
            t.mul_(1.0 / n)
    return t


class BucketedAverager(object):
    """The tower mean of average_gradients, per LAYER and overlapped with the backward pass.

    The gradients of recurrent layer l are complete as soon as its weight-gradient GEMMs have run (side lane 1,
    LSTMLayer.grad_event) -- long before the BPTT kernels of the layers below have finished, and those leave ~240 of
    the 256 CUs and all of xGMI idle.  So the layer's bucket (its contiguous run of the flat gradient buffer: 2.6 MB at
    5x256, 21 MB at 5x512) is clipped per variable (train_ctc.py:116: clip BEFORE the mean) and all-reduced on a
    communication stream right then, top layer first; what is not an encoder layer (heads, VGG front-end, attention
    decoder) forms the remaining bucket(s), reduced after the backward pass.  Same arithmetic as the single-bucket
    path (clip_by_norm is per variable, the mean is elementwise), same collective order on every rank.
    Not used with weight decay (its gradient term is added over the whole buffer after the backward pass)."""

    def __init__(self, model):
        self.model = model
        st = model.store
        enc = getattr(model, 'encoder', None)
        self.enc = enc if hasattr(enc, 'grad_ready_hook') else None
        layers = list(getattr(self.enc, 'layers', None) or [])
        self.layers = layers
        self.buckets = []          # one per GROUP of consecutive layers, top group first
        self.trigger = {}          # lowest layer index of a group -> index into self.buckets
        self.ok = bool(layers) and not float(getattr(model, 'weight_decay', 0.0) or 0.0) > 0.0
        covered = []
        runs = []
        if self.ok:
            for layer in layers:
                names = layer.var_names()
                idx = [st._index[n] for n in names]
                if idx != list(range(idx[0], idx[0] + len(idx))):
                    self.ok = False                       # not one contiguous run: keep the single bucket
                    break
                runs.append((names[0], names[-1], idx[0], idx[-1]))
            if self.ok and any(runs[i][3] + 1 != runs[i + 1][2] for i in range(len(runs) - 1)):
                self.ok = False                           # layers not back to back in the flat buffer
        if self.ok:
            # small layers share a collective: an all-reduce below a few MB is latency-bound on xGMI (ring set-up + one
            # launch per call, each also a few CUs beside the recurrence), so consecutive layers -- top down, the order
            # the backward pass finishes them -- are coalesced until the group holds ASR_DP_BUCKET_MB (default 6 MB:
            # three layers at 5x256, every layer alone at 5x512); the group goes out when its LOWEST layer is ready
            self.bucket_min_bytes = int(float(os.environ.get('ASR_DP_BUCKET_MB', '6')) * (1 << 20))
            hi = len(runs) - 1
            while hi >= 0:
                lo, nbytes = hi, 0
                while True:
                    nbytes += 4 * int(st.offsets_host[runs[lo][3] + 1] - st.offsets_host[runs[lo][2]])
                    if nbytes >= self.bucket_min_bytes or lo == 0:
                        break
                    lo -= 1
                self.trigger[lo] = len(self.buckets)
                b = st.bucket(runs[lo][0], runs[hi][1])
                b['layers'] = (lo, hi)
                self.buckets.append(b)
                covered.append((runs[lo][2], runs[hi][3]))
                hi = lo - 1
        self.rest = []
        if self.ok:
            taken = set()
            for a, b in covered:
                taken.update(range(a, b + 1))
            run = []
            for i, n in enumerate(st.names + [None]):
                if n is not None and i not in taken:
                    run.append(n)
                elif run:
                    self.rest.append(st.bucket(run[0], run[-1]))
                    run = []
        self.comm_stream = None
        self._seen = set()

    def enabled(self):
        return self.ok and (is_distributed() or bool(getattr(self, 'force', False)))

    def _reduce(self, bucket):
        clip = self.model.clip_grad_norm
        if clip is not None:
            ops.clip_by_norm_multi(bucket['grad'], bucket['plan'], float(clip))
        _allreduce_mean_(bucket['grad'])

    def _comm(self):
        dev = self.model.store.flat.device
        if dev.type != 'cuda':
            return None
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=dev)
        return self.comm_stream

    def begin(self):
        self._seen = set()
        if self.enc is not None:
            self.enc.grad_ready_hook = self.layer_ready

    def layer_ready(self, li, layer):
        """Hook of encoder.backward(): layer li's gradient GEMMs have been issued."""
        bi = self.trigger.get(li)
        if bi is None:
            return                       # a higher layer of a coalesced group: goes out with the group's lowest layer
        self._seen.add(bi)
        cs = self._comm()
        if cs is None:
            self._reduce(self.buckets[bi])
            return
        with torch.cuda.stream(cs):
            # side lane 1 is in order: the lowest layer's event implies the gradients of the layers above it
            ops.wait_event(layer.grad_event)
            self._reduce(self.buckets[bi])

    def finish(self):
        """After the backward pass (or for a rank with an empty shard, whose gradients are zero): the layers the hook
        has not seen -- same order as the backward pass issues them -- then the remaining buckets; the launch stream
        then waits for the communication stream."""
        if self.enc is not None:
            self.enc.grad_ready_hook = None
        cs = self._comm()
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(cs.device))
        ctx = torch.cuda.stream(cs) if cs is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            for bi in range(len(self.buckets)):          # top group first, as the hook issues them
                if bi not in self._seen:
                    self._reduce(self.buckets[bi])
            for b in self.rest:
                self._reduce(b)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        if cs is not None:
            torch.cuda.current_stream(cs.device).wait_stream(cs)


def averager_for(model):
    a = getattr(model, '_dp_averager', None)
    if a is None or a.model.store is not model.store:
        a = model._dp_averager = BucketedAverager(model)
    return a


def clip_and_average(model, optimizer, loss):
    """compute_gradients -> per-variable clip -> mean over towers for this rank's tower (loss None: empty shard)."""
    avg = averager_for(model)
    if avg.enabled():
        avg.begin()
        if loss is not None:
            optimizer.compute_gradients(loss, model=model)     # fires layer_ready per finished layer
        else:
            model.store.grad.zero_()
        avg.finish()
        return
    if loss is not None:
        gv = optimizer.compute_gradients(loss, model=model)
        if model.clip_grad_norm is not None:
            model._clip_gradients(gv)
    else:
        model.store.grad.zero_()
    average_gradients(model.store)


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
        # training collectives keep a short watchdog: a rank that dies (ErrorWatch / AsrError, any exception) must not
        # leave the others blocked in the per-layer all-reduces for hours.  The one long wait of the recipes -- rank 0
        # alone evaluates dev / test sets at epoch ends (examples/librispeech/training/train_ctc.py) -- goes through
        # broadcast_decision(), which has its own long-timeout group
        import datetime
        timeout = datetime.timedelta(minutes=float(os.environ.get('ASR_DIST_TIMEOUT_MINUTES', '10')))
        if device.type == 'cuda':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device, timeout=timeout)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=timeout)
        if device.type == 'cuda':
            warm_up_collectives(device)
    return rank, world


def warm_up_collectives(device):
    """RCCL bootstrap (ring / tree setup over xGMI, the C ABI's own communicator) and one tiny all-reduce on each path
    NOW, at start-up, instead of inside the first training step -- where its seconds would land in step 1 of every
    rank and, with ranks bootstrapping at different speeds, in the watchdog."""
    if not is_distributed():
        return
    device = torch.device(device)
    t = torch.ones(256, dtype=torch.float32, device=device)
    dist.all_reduce(t)
    comm = native_comm(device)
    if comm is not None:
        comm.allreduce_mean(t)
    if device.type == 'cuda':
        torch.cuda.synchronize(device)


_decision_group = None


def broadcast_decision(value, src=0):
    """rank `src`'s Python value (early-stop flag, new learning rate, run directory) to every rank.  The other ranks may
    wait here for as long as rank 0 needs to evaluate the dev / test sets, so this uses its own gloo group with a long
    timeout (ASR_DIST_EVAL_TIMEOUT_HOURS, default 6) and leaves the training collectives' watchdog short."""
    global _decision_group
    if not is_distributed():
        return value
    if _decision_group is None:
        import datetime
        import os
        hours = float(os.environ.get('ASR_DIST_EVAL_TIMEOUT_HOURS', '6'))
        _decision_group = dist.new_group(backend='gloo', timeout=datetime.timedelta(hours=hours))
    box = [value]
    dist.broadcast_object_list(box, src=src, group=_decision_group)
    return box[0]


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
    loss = None
    if B > 0:
        loss, logits = model.compute_loss(inputs, labels, inputs_seq_len, keep_prob)
    clip_and_average(model, optimizer, loss)
    loss = loss.detach() if loss is not None else torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
    optimizer.apply_gradients(None, learning_rate=learning_rate)
    return average_scalar(loss), logits


def tower_step_with(model, optimizer, loss_fn, learning_rate=None):
    """tower_step for any model family: `loss_fn()` runs this rank's forward and returns the loss tensor of
    compute_loss (or None when the rank's shard of the global batch is empty); the rest is the tower loop of
    examples/csj/training/train_attention.py:90-150 -- gradients, per-variable clip on the tower, mean over towers,
    identical update.  Returns the loss averaged over towers."""
    loss = loss_fn()
    clip_and_average(model, optimizer, loss)
    loss = loss.detach() if loss is not None else torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
    optimizer.apply_gradients(None, learning_rate=learning_rate)
    return average_scalar(loss)
