#!/usr/bin/env python
"""Headline benchmark: acoustic frames/sec of BLSTM-CTC training on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d cfg B): TIMIT-61-shaped synthetic batch,
5x256 BLSTM-CTC, bf16 MFMA operands / fp32 state, B=16 utterances per GPU, D=120 (40 log-mel
x {static, delta, delta-delta}), C=62, seq_len ~ U{100..778}, L = clip(len//8, 5, 75);
recipe hyper-parameters of the repo's own config (blstm_ctc_100h_char.yml): rmsprop 1e-3,
clip_grad_norm 5, clip_activation 50, dropout 0.2.
A step = forward + CTC loss + backward + per-variable clip + (all-reduce) + optimizer update.

The ONE JSON line rank 0 prints holds
  value / ms_per_step    K steps bracketed by barrier + synchronize, inputs resident in HBM (the contract's number);
  step_ms                per-step durations from HIP events on the launch stream: median / min / max;
  h2d_inclusive          the same K steps with the batch uploaded from pinned host memory every step
                         (double-buffered on a copy stream, SURVEY 8d's step definition) -- reported, never `value`;
  parity                 CTC-loss match and greedy-label match against the CPU oracle on the same batch cut to its
                         first --cpu-tmax frames (computed outside the timed region, before training starts);
  roofline               dominant kernel, HIP-event duration measured live in the timed region;
  cpu_baseline           oracle/fast_cpu.py port of the TF1 CPU path on a bounded sample;
  cfgA                   BASELINE configs[0] (TIMIT-39, 2x128, fp32) timed the same way: the configuration the
                         1e-4 fp32 loss tolerance of north_star is a statement about (N = 1 only).
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


def truncate_batch(x, seq_len, labels, tcut):
    """The same utterances cut to their first tcut frames, labels cut to stay feasible (bounded CPU sample)."""
    sl = np.minimum(seq_len, tcut).astype(np.int32)
    xc = x[:, :int(sl.max())].copy()
    for b in range(len(sl)):
        xc[b, sl[b]:] = 0
    labs = [list(l[:max(1, int(n) // 8)]) for l, n in zip(labels, sl)]
    dense = np.full((len(labs), max(len(l) for l in labs)), -1, dtype=np.int64)
    for b, l in enumerate(labs):
        dense[b, :len(l)] = l
    return xc, sl, labs, dense


class KernelTimer(object):
    """HIP-event brackets (torch.cuda.Event on the current stream == the launch stream of ops.*)."""

    def __init__(self, ops, names):
        self.ops, self.names = ops, names
        self.records = {n: [] for n in names}
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

    def uninstall(self):
        for n, o in self._orig.items():
            setattr(self.ops, n, o)

    def summary(self):
        out = {}
        for n, evs in self.records.items():
            if evs:
                ms = [a.elapsed_time(b) for a, b in evs]
                out[n] = dict(calls=len(ms), total_ms=float(np.sum(ms)), avg_us=float(np.mean(ms) * 1e3))
        return out


def parity_vs_oracle(model, x, seq_len, labels, dense, L, dtype, tcut):
    """Device loss / greedy labels against the CPU oracle on the batch cut to its first tcut frames.  For the bf16
    operand path the oracle is evaluated on the bf16-rounded operands (inputs, kernels, emitted h), so what is compared
    is the arithmetic, not the precision choice; for fp32 it is the plain fp64 oracle (tolerance 1e-4, north_star)."""
    from oracle import decoders as odec
    from oracle import lstm as olstm
    from oracle import model as omodel
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import sparsetensor2list
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    xc, sl, labs, dn = truncate_batch(x, seq_len, labels, tcut)
    t0 = time.perf_counter()
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ref = omodel.ctc_model_forward(sd, xc, labs, sl, L, cell_clip=50.0, want_grads=False,
                                   operand_round=olstm.bf16_round_t if dtype == 'bf16' else None)
    loss, logits = model.compute_loss(xc, dn, sl, keep_prob=1.0, is_training=False)
    B = len(sl)
    C = logits.shape[2]
    hyp = [list(h) for h in sparsetensor2list(model.decoder(logits, sl, beam_width=1), B)]
    ref_hyp = odec.greedy_decode(np.transpose(ref['logits'], (1, 0, 2)), sl, C - 1)
    per_utt = np.abs(model.ctc_losses.cpu().numpy() - ref['ctc_losses']) / np.abs(ref['ctc_losses'])
    return dict(loss_device=float(loss.item()), loss_oracle=float(ref['total_loss']),
                loss_rel_err_vs_oracle=abs(float(loss.item()) - ref['total_loss']) / abs(ref['total_loss']),
                per_utterance_loss_rel_err_max=float(per_utt.max()),
                greedy_label_mismatch=int(sum(h != r for h, r in zip(hyp, ref_hyp))),
                greedy_labels_compared=int(sum(len(r) for r in ref_hyp)), utterances=B,
                oracle='oracle.model fp64' + (' on bf16-rounded operands' if dtype == 'bf16' else ''),
                sample='first %d frames of every utterance of the timed batch (%d valid frames), dropout off'
                       % (tcut, int(sl.sum())), seconds=time.perf_counter() - t0)


def run_workload(args, wl, dev, world, rank, local_rank, want_parity, want_h2d):
    """Times one workload description `wl`; returns the result dict of this rank (rank 0 aggregates)."""
    import torch.distributed as dist
    from tensorflow_end2end_speech_recognition_amd import ops
    from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC
    from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu

    H, L, C = wl['units'], wl['layers'], wl['classes'] + 1
    # every rank gets its own shard of the global batch (weak scaling: B per GPU fixed)
    x, seq_len, labels, dense = make_batch(wl['seed'] + rank, wl['batch'], wl['input_size'], C, wl['tmin'], wl['tmax'])
    model = CTC('blstm', wl['input_size'], H, L, wl['classes'], parameter_init=0.1, clip_grad_norm=5.0,
                clip_activation=50, dtype=wl['dtype'], device=str(dev), seed=0)
    multi_gpu.broadcast_parameters(model.store)
    frames = int(seq_len.sum())
    res = dict(frames=frames)
    if want_parity and rank == 0:
        res['parity'] = parity_vs_oracle(model, x, seq_len, labels, dense, L, wl['dtype'], args.cpu_tmax)
    xd = torch.tensor(x, device=dev)
    sld = torch.tensor(seq_len, device=dev)
    opt = model._set_optimizer('rmsprop', 1e-3)

    # only the serial kernels are bracketed (11 launches/step): an event pair around each of the ~40
    # small GEMMs costs ~3.5 ms/step of queue serialisation and would distort the number being measured
    timer = KernelTimer(ops, ['lstm_fwd', 'lstm_bwd', 'ctc_loss'])
    timer.install()

    def step(xin):
        loss, logits = model.compute_loss(xin, dense, sld, keep_prob=wl['keep_prob'])
        # gradients -> per-variable clip on the tower (BEFORE averaging) -> mean over towers: per encoder layer on a
        # communication stream under the BPTT of the layers below when N > 1 (multi_gpu.BucketedAverager)
        multi_gpu.clip_and_average(model, opt, loss)
        opt.apply_gradients(None)
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step(xd)
    fence()
    K = args.steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host_issue = 0.0
    timer.enabled = True
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(K):
        h0 = time.perf_counter()
        loss = step(xd)
        marks[i + 1].record()
        host_issue += time.perf_counter() - h0
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    res['elapsed'] = elapsed
    res['final_loss'] = float(loss.item())
    res['handoff_flags'] = ops.check_async_errors(local_rank)   # sticky error word of the multi-CU recurrence kernels
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(K)])
    res['step_ms'] = dict(median=float(np.median(per_step)), min=float(per_step.min()), max=float(per_step.max()),
                          host_issue_mean=host_issue / K * 1e3,
                          note='HIP events between steps on the launch stream; host_issue = time the Python side needs '
                               'to enqueue one step (must stay below the GPU time or the step becomes host-bound)')
    res['kernels'] = timer.summary()

    if want_h2d:
        # the same K steps with the batch coming from pinned host memory each step: upload of step i+1 on a copy
        # stream under the compute of step i (what a prefetching input pipeline does); all K uploads in the bracket
        xp = torch.from_numpy(x).pin_memory()
        bufs = [torch.empty_like(xd), torch.empty_like(xd)]
        copy_stream = torch.cuda.Stream(device=dev)
        ready = [None, None]
        done = [None, None]

        def upload(i):
            with torch.cuda.stream(copy_stream):
                if done[i % 2] is not None:
                    copy_stream.wait_event(done[i % 2])        # the step that last read this buffer has finished
                bufs[i % 2].copy_(xp, non_blocking=True)
                e = torch.cuda.Event()
                e.record(copy_stream)
                ready[i % 2] = e
        fence()
        t0 = time.perf_counter()
        upload(0)
        for i in range(K):
            if i + 1 < K:
                upload(i + 1)
            torch.cuda.current_stream().wait_event(ready[i % 2])
            loss = step(bufs[i % 2])
            e = torch.cuda.Event()
            e.record()
            done[i % 2] = e
        fence()
        res['elapsed_h2d'] = time.perf_counter() - t0
    timer.uninstall()
    res['x'], res['seq_len'], res['labels'], res['model'] = x, seq_len, labels, model
    return res


def aggregate(res, args, world, dev):
    import torch.distributed as dist
    out = {}
    for key in ('elapsed', 'elapsed_h2d'):
        if key not in res:
            continue
        t = torch.tensor([res[key]], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[key] = float(t.item())
    f = torch.tensor([float(res['frames'])], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(f, op=dist.ReduceOp.SUM)
    out['total_frames'] = float(f.item())
    return out


def roofline(wl, frames, ks, traffic_table):
    H, L = wl['units'], wl['layers']
    dom = max(('lstm_fwd', 'lstm_bwd', 'ctc_loss'), key=lambda n: ks.get(n, {}).get('total_ms', 0))
    s_act = 2 if wl['dtype'] == 'bf16' else 4
    # algorithmic HBM bytes per launch of the recurrence kernels (DESIGN.md "Kernels"):
    #   fwd: read x W_x+b 16H, write gates s*4H + c 4H + h s*H             per valid frame per direction
    #   bwd: read gates s*4H + c 4H + dh 4H, write dgates s*4H              per valid frame per direction
    per_frame = {'lstm_fwd': 20 * H + 5 * s_act * H, 'lstm_bwd': 8 * H + 8 * s_act * H}
    if dom not in per_frame:
        return None
    flops_frame = 2 * 4 * H * H   # recurrent h W_h (fwd) / dG W_h^T (bwd), per frame per direction
    bytes_launch = frames * 2 * per_frame[dom] + 2 * 4 * H * H * s_act
    dur = ks[dom]['avg_us'] * 1e-6
    ach = bytes_launch / dur / 1e9
    T = wl['tmax']
    traffic = None
    src = None
    key = '%dx%d_%s_B%d_T%d' % (L, H, wl['dtype'], wl['batch'], T)
    if traffic_table and key in traffic_table.get('workloads', {}):
        traffic = traffic_table['workloads'][key].get(dom)
        src = traffic_table.get('source')
    peak_tf = MFMA_BF16_PEAK_TF if wl['dtype'] == 'bf16' else MFMA_F32_PEAK_TF
    return dict(kernel=dom, bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS,
                traffic=traffic, traffic_source=src, avg_launch_us=ks[dom]['avg_us'],
                us_per_recurrence_step=ks[dom]['avg_us'] / T, algorithmic_bytes_per_launch=bytes_launch,
                mfma_tflops=frames * 2 * flops_frame / dur / 1e12,
                mfma_frac=frames * 2 * flops_frame / dur / 1e12 / peak_tf,
                note='serial recurrence over T frames on one 16-utterance MFMA tile per direction: bound by the '
                     'per-step chain (LDS operand reads, gate math issue, one cross-CU L2 hop), not by HBM or MFMA '
                     'throughput (DESIGN.md section 4); traffic = HBM bytes per launch from a separate rocprofv3 --pmc '
                     'pass of this command (cannot be collected inside the timed run), null if no pass matches')


def cpu_baseline(args, wl, res):
    from oracle import fast_cpu
    model, x, seq_len, labels = res['model'], res['x'], res['seq_len'], res['labels']
    sd = {k: v.cpu().numpy() for k, v in model.store.state_dict().items()}
    ncores = min(os.cpu_count() or 1, args.cpu_threads)
    xc, sl, labs, _ = truncate_batch(x, seq_len, labels, args.cpu_tmax)
    cm = fast_cpu.CpuBLSTMCTC(sd, wl['layers'], cell_clip=50.0, clip_grad_norm=5.0, threads=ncores,
                              optimizer='rmsprop')
    t_cpu = fast_cpu.time_train_steps(cm, xc, labs, sl, steps=args.cpu_steps, warmup=1)
    cframes = int(sl.sum())
    return dict(value=cframes / t_cpu, unit='frames/s', cores=ncores, kind='port',
                sample='%d timed training step(s) after 1 warm-up step of the same %d-utterance batch cut to its first '
                       '%d frames (%d valid frames), rmsprop, torch-CPU fp32 restatement of the TF1 path '
                       '(oracle/fast_cpu.py) on %d threads (torch.set_num_threads) of a %d-core host'
                       % (args.cpu_steps, wl['batch'], args.cpu_tmax, cframes, ncores, os.cpu_count() or 1),
                seconds_per_step=t_cpu)


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
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-cfgA', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--cpu-tmax', type=int, default=256,
                    help='CPU legs (baseline, oracle parity): the same batch truncated to its first N frames')
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: libraries that print banners to file descriptor 1 (RCCL's
    # version block, gloo's rank messages) are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d' % (args.gpus, args.gpus))
    # dry-run knobs for a box with fewer GPUs than ranks (scripts/r02_dp_dryrun.sh): ASR_BENCH_DEVICE pins every rank to
    # one device, ASR_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    dev_index = int(os.environ.get('ASR_BENCH_DEVICE', local_rank))
    backend = os.environ.get('ASR_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = dict(units=args.units, layers=args.layers, classes=args.classes, dtype=args.dtype, batch=args.batch,
              input_size=args.input_size, tmin=args.tmin, tmax=args.tmax, keep_prob=args.keep_prob, seed=1)
    res = run_workload(args, wl, dev, world, rank, dev_index, want_parity=not args.no_parity, want_h2d=True)
    agg = aggregate(res, args, world, dev)
    value = agg['total_frames'] * args.steps / agg['elapsed']

    if rank == 0:
        traffic_table = None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_hbm_traffic.json')
        if os.path.exists(tpath):
            traffic_table = json.load(open(tpath))
        ks = res['kernels']
        H, L, C = wl['units'], wl['layers'], wl['classes'] + 1
        out = dict(metric='acoustic frames/sec (train), TIMIT-shaped BLSTM-CTC', value=value, unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=agg['elapsed'] / args.steps * 1e3, higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype=args.dtype, data='synthetic',
                   config=dict(workload='TIMIT 61-phone %dx%d BLSTM-CTC, B=%d/GPU, D=%d, C=%d, '
                                        'seq_len~U{%d..%d}, dropout %.1f, rmsprop, train step'
                                        % (L, H, args.batch, args.input_size, C, args.tmin, args.tmax,
                                           1 - args.keep_prob),
                               global_batch=args.batch * world, frames_per_step=agg['total_frames'],
                               parallelism='dp%d' % world),
                   step_ms=res['step_ms'],
                   h2d_inclusive=dict(value=agg['total_frames'] * args.steps / agg['elapsed_h2d'], unit='frames/s',
                                      ms_per_step=agg['elapsed_h2d'] / args.steps * 1e3,
                                      note='batch uploaded from pinned host memory every step, double-buffered on a '
                                           'copy stream; reported next to `value`, which has the inputs resident'),
                   final_loss=res['final_loss'], cluster_handoff_flags=res['handoff_flags'],
                   parity=res.get('parity'), kernels=ks, roofline=roofline(wl, res['frames'], ks, traffic_table),
                   cpu_baseline=None, cfgA=None)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, wl, res)
        del res
        if world == 1 and not args.no_cfgA:
            # BASELINE configs[0]: TIMIT-39, 2x128 BLSTM-CTC, fp32 (exact fp32 MFMA path), B=16, dropout 0.5
            wa = dict(units=128, layers=2, classes=39, dtype='f32', batch=16, input_size=120, tmin=100, tmax=778,
                      keep_prob=0.5, seed=0)
            ra = run_workload(args, wa, dev, world, rank, dev_index, want_parity=not args.no_parity, want_h2d=False)
            out['cfgA'] = dict(workload='TIMIT 39-phone 2x128 BLSTM-CTC fp32, B=16, D=120, C=40, seq_len~U{100..778}, '
                                        'dropout 0.5, rmsprop, train step',
                               value=ra['frames'] * args.steps / ra['elapsed'], unit='frames/s', dtype='f32',
                               ms_per_step=ra['elapsed'] / args.steps * 1e3, step_ms=ra['step_ms'],
                               final_loss=ra['final_loss'], parity=ra.get('parity'), kernels=ra['kernels'],
                               cpu_baseline=None if args.no_cpu_baseline else cpu_baseline(args, wa, ra))
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    if world > 1:
        dist.destroy_process_group()
    os.close(result_fd)


if __name__ == '__main__':
    main()
