#!/usr/bin/env python
"""Headline benchmark: acoustic frames/sec of BLSTM-CTC training on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d cfg B): TIMIT-61-shaped synthetic batch,
5x256 BLSTM-CTC, bf16 MFMA operands / fp32 state, B=16 utterances per GPU, D=120 (40 log-mel
x {static, delta, delta-delta}), C=62, seq_len ~ U{100..778}, L = clip(len//8, 5, 75);
recipe hyper-parameters of the repo's own config (blstm_ctc_100h_char.yml): rmsprop 1e-3,
clip_grad_norm 5, clip_activation 50, dropout 0.2.
A step = forward + CTC loss + backward + per-variable clip + (all-reduce) + optimizer update,
inputs already resident in HBM.  value = valid frames (sum of seq_len over all ranks) x K / time.
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, measured with HIP events on
the launch stream inside the timed region) and `cpu_baseline` (oracle/fast_cpu.py port).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16
MFMA_F32_PEAK_TF = 157.3


def make_batch(seed, B, D, C, tmin, tmax):
    rng = np.random.RandomState(seed)
    seq_len = rng.randint(tmin, tmax + 1, size=B).astype(np.int32)
    T = int(seq_len.max())
    x = rng.randn(B, T, D).astype(np.float32)
    labels = []
    for b in range(B):
        x[b, seq_len[b]:] = 0
        L = int(np.clip(seq_len[b] // 8, 5, 75))
        labels.append(rng.randint(0, C - 1, size=L).tolist())
    Lmax = max(len(l) for l in labels)
    dense = np.full((B, Lmax), -1, dtype=np.int64)
    for b, l in enumerate(labels):
        dense[b, :len(l)] = l
    return x, seq_len, labels, dense


class KernelTimer(object):
    """HIP-event brackets (torch.cuda.Event on the current stream == the launch stream of ops.*)."""

    def __init__(self, ops, names):
        self.ops, self.names = ops, names
        self.records = {n: [] for n in names}
        self.meta = {n: [] for n in names}
        self.enabled = False
        self._orig = {}

    def install(self):
        for n in self.names:
            orig = getattr(self.ops, n)
            self._orig[n] = orig

            def wrapped(*a, _n=n, _o=orig, **k):
                if not self.enabled:
                    return _o(*a, **k)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _o(*a, **k)
                e1.record()
                self.records[_n].append((e0, e1))
                return r
            setattr(self.ops, n, wrapped)

    def summary(self):
        out = {}
        for n, evs in self.records.items():
            if evs:
                ms = [a.elapsed_time(b) for a, b in evs]
                out[n] = dict(calls=len(ms), total_ms=float(np.sum(ms)), avg_us=float(np.mean(ms) * 1e3))
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--units', type=int, default=256)
    ap.add_argument('--layers', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='utterances per GPU')
    ap.add_argument('--classes', type=int, default=61)
    ap.add_argument('--input-size', type=int, default=120)
    ap.add_argument('--tmin', type=int, default=100)
    ap.add_argument('--tmax', type=int, default=778)
    ap.add_argument('--keep-prob', type=float, default=0.8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--cpu-tmax', type=int, default=256,
                    help='CPU baseline sample: the same batch truncated to its first N frames')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d' % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu

    H, L, C = args.units, args.layers, args.classes + 1
    # every rank gets its own shard of the global batch (weak scaling: B per GPU fixed)
    x, seq_len, labels, dense = make_batch(1 + rank, args.batch, args.input_size, C, args.tmin, args.tmax)
    model = CTC('blstm', args.input_size, H, L, args.classes, parameter_init=0.1, clip_grad_norm=5.0,
                clip_activation=50, dtype=args.dtype, device=str(dev), seed=0)
    multi_gpu.broadcast_parameters(model.store)
    xd = torch.tensor(x, device=dev)
    sld = torch.tensor(seq_len, device=dev)
    opt = model._set_optimizer('rmsprop', 1e-3)
    frames = int(seq_len.sum())

    # only the serial kernels are bracketed (11 launches/step): an event pair around each of the ~40
    # small GEMMs costs ~3.5 ms/step of queue serialisation and would distort the number being measured
    timer = KernelTimer(ops, ['lstm_fwd', 'lstm_bwd', 'ctc_loss'])
    timer.install()

    def step():
        loss, logits = model.compute_loss(xd, dense, sld, keep_prob=args.keep_prob)
        gv = opt.compute_gradients(loss, model=model)
        model._clip_gradients(gv)                       # clip per tower BEFORE averaging
        multi_gpu.average_gradients(model.store)        # RCCL all-reduce / N
        opt.apply_gradients(gv)
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    final_loss = float(loss.item())
    handoff_flags = ops.check_async_errors(local_rank)    # sticky error word of the multi-CU recurrence kernels

    tot_frames = torch.tensor([float(frames)], device=dev)
    el = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(tot_frames, op=dist.ReduceOp.SUM)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    total_frames = float(tot_frames.item())
    value = total_frames * args.steps / elapsed

    if rank == 0:
        ks = timer.summary()
        dom = max(('lstm_fwd', 'lstm_bwd', 'ctc_loss'), key=lambda n: ks.get(n, {}).get('total_ms', 0))
        s_act = 2 if args.dtype == 'bf16' else 4
        # algorithmic HBM bytes per launch of the recurrence kernels (DESIGN.md "Kernels"):
        #   fwd: read x W_x+b 16H, write gates s*4H + c 4H + h s*H             per valid frame per direction
        #   bwd: read gates s*4H + c 4H + dh 4H, write dgates s*4H              per valid frame per direction
        per_frame = {'lstm_fwd': 20 * H + 5 * s_act * H, 'lstm_bwd': 8 * H + 8 * s_act * H}
        # measured HBM bytes per launch for THIS default workload (profiles/r01b_pmc_hbm.md:
        # 2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, gfx950 correction applied; includes the
        # per-step cross-CU exchange of the cluster kernels); null otherwise
        default_cfg = (args.units, args.layers, args.batch, args.tmax, args.dtype) == (256, 5, 16, 778, 'bf16')
        measured_traffic = {'lstm_bwd': 144.5e6, 'lstm_fwd': 180.1e6}
        flops_frame = 2 * 4 * H * H   # recurrent h W_h (fwd) / dG W_h^T (bwd), per frame per direction
        roof = None
        if dom in per_frame:
            launches_per_step = L
            bytes_launch = frames * 2 * per_frame[dom] + 2 * 4 * H * H * s_act
            dur = ks[dom]['avg_us'] * 1e-6
            ach = bytes_launch / dur / 1e9
            roof = dict(kernel=dom, bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=ach / HBM_PEAK_GBS, traffic=(measured_traffic[dom] if default_cfg else None),
                        avg_launch_us=ks[dom]['avg_us'],
                        algorithmic_bytes_per_launch=bytes_launch,
                        mfma_tflops=frames * 2 * flops_frame / dur / 1e12,
                        mfma_frac=frames * 2 * flops_frame / dur / 1e12 /
                        (MFMA_BF16_PEAK_TF if args.dtype == 'bf16' else MFMA_F32_PEAK_TF),
                        note='serial recurrence over T frames on one 16-utterance MFMA tile per direction: '
                             'bound by the per-step chain (LDS operand reads, gate math issue, one cross-CU '
                             'L2 hop), not by HBM or MFMA throughput (DESIGN.md section 4)')
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import fast_cpu
            sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
            ncores = min(os.cpu_count() or 1, args.cpu_threads)
            # bounded sample (10-30 s of CPU work): the same utterances cut to their first cpu_tmax frames
            # (per-frame cost of the recurrence does not depend on T), labels cut to stay feasible
            tcut = args.cpu_tmax
            sl_c = np.minimum(seq_len, tcut)
            x_c = x[:, :int(sl_c.max())].copy()
            lab_c = [l[:max(1, int(n) // 8)] for l, n in zip(labels, sl_c)]
            cm = fast_cpu.CpuBLSTMCTC(sd, L, cell_clip=50.0, clip_grad_norm=5.0, threads=ncores)
            t_cpu = fast_cpu.time_train_steps(cm, x_c, lab_c, sl_c, steps=args.cpu_steps, warmup=0)
            cframes = int(sl_c.sum())
            cpu = dict(value=cframes / t_cpu, unit='frames/s', cores=ncores, kind='port',
                       sample='%d training step(s) of the same %d-utterance batch truncated to its first %d '
                              'frames (%d valid frames), torch-CPU fp32 restatement of the TF1 path '
                              '(oracle/fast_cpu.py), %d threads of a %d-core host'
                              % (args.cpu_steps, args.batch, tcut, cframes, ncores, os.cpu_count() or 1),
                       seconds_per_step=t_cpu)
        out = dict(metric='acoustic frames/sec (train), TIMIT-shaped BLSTM-CTC', value=value, unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype=args.dtype, data='synthetic',
                   config=dict(workload='TIMIT 61-phone %dx%d BLSTM-CTC, B=%d/GPU, D=%d, C=%d, '
                                        'seq_len~U{%d..%d}, dropout %.1f, rmsprop, train step'
                                        % (L, H, args.batch, args.input_size, C, args.tmin, args.tmax,
                                           1 - args.keep_prob),
                               global_batch=args.batch * world, frames_per_step=total_frames,
                               parallelism='dp%d' % world),
                   final_loss=final_loss, cluster_handoff_flags=handoff_flags, kernels=ks, roofline=roof,
                   cpu_baseline=cpu)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
