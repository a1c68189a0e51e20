#!/usr/bin/env python
"""Generate golden vectors from the REFERENCE'S OWN code (run in the build
container only: needs /root/reference; the .npz outputs are committed and are
what travels to the GPU box).

    python tests/golden/make_golden.py

Produces
  decoders_v1.npz : seeded posteriors + outputs of
      /root/reference/models/ctc/decoders/greedy_decoder.py  GreedyDecoder
      /root/reference/models/ctc/decoders/beam_search_decoder.py BeamSearchDecoder
    called with B=1 slices (as the reference itself does at
    examples/librispeech/metrics/ctc.py:220-223; B>1 ragged np.array raises on
    numpy >= 1.24).  Posteriors are float64 so the reference arithmetic is
    float64 under both old (value-based) and NEP-50 numpy promotion rules.
  splice_v1.npz   : inputs/outputs of utils/io/inputs/splicing.py do_splice and
      utils/io/inputs/frame_stacking.py stack_frame.
  labels_v1.json  : outputs of the label maps (utils/io/labels/*.py), Map2phone39 and compute_wer.
  controller_v1.json : learning-rate trajectories of utils/training/learning_rate_controller.py Controller.
  sparsetensor_v1.json : list2sparsetensor / sparsetensor2list of utils/io/labels/sparsetensor.py.
  datasets_v1.npz : every batch the reference's four DatasetBase iterators (utils/dataset/*.py) yield on a small
      generated corpus (numpy < 1.24 ragged-array semantics restored by a shim, see _OldNumpy).
"""
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def make_decoders():
    sys.path.insert(0, REF)
    from models.ctc.decoders.greedy_decoder import GreedyDecoder
    from models.ctc.decoders.beam_search_decoder import BeamSearchDecoder
    warnings.simplefilter('ignore')
    rng = np.random.RandomState(1234)
    cases = []
    # (T, C, beam widths, sharpness): peaky posteriors resemble trained CTC output
    specs = [(12, 5, (1, 2, 4), 1.0), (30, 8, (1, 3, 8), 2.0), (50, 29, (1, 5, 20), 3.0),
             (40, 40, (4, 20), 4.0), (25, 62, (10,), 3.0), (16, 6, (20,), 0.3),
             (1, 4, (1, 3), 1.0), (60, 29, (20,), 6.0)]
    out = {}
    n = 0
    for (T, C, widths, sharp) in specs:
        for rep in range(3):
            logits = rng.randn(1, T, C) * sharp
            # make blank (C-1) frequent like a trained model, and add repeats
            logits[0, :, C - 1] += sharp * rng.rand()
            for t in range(1, T):
                if rng.rand() < 0.35:
                    logits[0, t] = logits[0, t - 1] + 0.1 * rng.randn(C)
            probs = _softmax(logits).astype(np.float64)
            seq_len = np.array([T if rep != 2 else max(1, T - 3)], dtype=np.int32)
            g = GreedyDecoder(blank_index=C - 1)(probs, seq_len)
            out['c%d_probs' % n] = probs
            out['c%d_seq_len' % n] = seq_len
            out['c%d_greedy' % n] = np.asarray(g[0], dtype=np.int64)
            for w in widths:
                hyp, score = BeamSearchDecoder(space_index=-1, blank_index=C - 1)(
                    probs, seq_len, beam_width=w)
                out['c%d_beam%d' % (n, w)] = np.asarray(hyp[0], dtype=np.int64)
                out['c%d_beam%d_score' % (n, w)] = np.asarray(score[0], dtype=np.float64)
            out['c%d_widths' % n] = np.array(widths, dtype=np.int64)
            n += 1
    out['num_cases'] = np.array(n)
    np.savez_compressed(os.path.join(HERE, 'decoders_v1.npz'), **out)
    print('decoders_v1.npz: %d cases' % n)


def make_splice():
    sys.path.insert(0, REF)
    # utils/io/inputs/frame_stacking.py imports utils.progressbar -> tqdm; fine here
    from utils.io.inputs.splicing import do_splice
    from utils.io.inputs.frame_stacking import stack_frame
    rng = np.random.RandomState(4321)
    out = {}
    n = 0
    for (T, ch, splice, num_stack) in [(7, 2, 3, 1), (20, 4, 5, 1), (13, 3, 11, 1), (9, 2, 3, 2),
                                       (3, 2, 5, 1), (30, 40, 11, 1)]:
        x = rng.randn(2, T, ch * 3 * num_stack)
        y = do_splice(x, splice=splice, batch_size=2, num_stack=num_stack)
        out['s%d_in' % n] = x
        out['s%d_out' % n] = y
        out['s%d_cfg' % n] = np.array([splice, num_stack])
        n += 1
    out['num_splice'] = np.array(n)
    m = 0
    for (T, D, num_stack, num_skip) in [(10, 6, 2, 2), (11, 6, 3, 3), (7, 3, 3, 2), (5, 4, 2, 1),
                                        (1, 3, 2, 2), (23, 120, 2, 2)]:
        x = rng.randn(T, D)
        y = stack_frame(np.array([x]), num_stack, num_skip)
        out['f%d_in' % m] = x
        out['f%d_out' % m] = np.asarray(y[0])
        out['f%d_cfg' % m] = np.array([num_stack, num_skip])
        m += 1
    out['num_stack'] = np.array(m)
    np.savez_compressed(os.path.join(HERE, 'splice_v1.npz'), **out)
    print('splice_v1.npz: %d splice + %d stack cases' % (n, m))


def make_labels():
    """Outputs of the reference's label / scoring utilities (utils/io/labels/{phone,character,word}.py,
    examples/timit/metrics/mapping.py Map2phone39, utils/evaluation/edit_distance.py compute_wer) run on its
    own mapping files (examples/timit/metrics/mapping_files/*.txt).  Only OUTPUTS are stored.
    python-Levenshtein and TensorFlow are not installed here, so the module is imported with empty stubs for
    both and only compute_wer (pure numpy) is recorded."""
    import json
    sys.path.insert(0, REF)
    sys.modules.setdefault('Levenshtein', types.ModuleType('Levenshtein'))
    sys.modules.setdefault('tensorflow', types.ModuleType('tensorflow'))   # only compute_edit_distance uses it
    from utils.io.labels.phone import Phone2idx, Idx2phone
    from utils.io.labels.character import Char2idx, Idx2char
    from utils.io.labels.word import Idx2word
    from utils.evaluation.edit_distance import compute_wer
    sys.path.insert(0, os.path.join(REF, 'examples', 'timit', 'metrics'))
    from mapping import Map2phone39
    mf = os.path.join(REF, 'examples', 'timit', 'metrics', 'mapping_files')
    rng = np.random.RandomState(77)
    out = {}
    for lt, n in (('phone61', 61), ('phone48', 48), ('phone39', 39)):
        i2p = Idx2phone(os.path.join(mf, lt + '.txt'))
        table = i2p(np.arange(n)).split(' ')
        out[lt + '_table'] = table                                   # idx -> phone (what the map file says)
        out[lt + '_file_tokens'] = [l.split()[0] for l in open(os.path.join(mf, lt + '.txt')) if l.strip()]
        p2i = Phone2idx(os.path.join(mf, lt + '.txt'))
        seqs = [rng.randint(0, n, size=rng.randint(1, 15)).tolist() + [-1] * rng.randint(0, 3) for _ in range(6)]
        out[lt + '_idx2phone'] = [[s_, i2p(np.array(s_))] for s_ in seqs]
        out[lt + '_phone2idx'] = [[table[i] for i in s_ if i >= 0] for s_ in seqs], \
            [p2i([table[i] for i in s_ if i >= 0]).tolist() for s_ in seqs]
        m39 = Map2phone39(lt, os.path.join(mf, 'phone2phone.txt'))
        out[lt + '_to39_table'] = [[ph, m39([ph])] for ph in table]  # per-phone folding ([] = dropped)
        out[lt + '_to39_seq'] = [[[table[i] for i in s_ if i >= 0], m39([table[i] for i in s_ if i >= 0])] for s_ in seqs]
    for name, kw in (('character', {}), ('character_capital_divide', dict(capital_divide=True, space_mark='_'))):
        path = os.path.join(mf, name + '.txt')
        rows = [l.strip().split() for l in open(path) if l.strip()]
        chars = [r[0] for r in rows]
        out[name + '_table'] = [[r[0], int(r[1])] for r in rows]
        i2c = Idx2char(path, **kw)
        seqs = [rng.randint(0, len(chars), size=rng.randint(1, 20)).tolist() + [-1] * rng.randint(0, 3) for _ in range(8)]
        out[name + '_idx2char'] = [[s_, i2c(np.array(s_))] for s_ in seqs]
        strs = [''.join(chars[i] for i in s_ if i >= 0) for s_ in seqs]
        singles = [c for c in chars if len(c) == 1]
        strs = [''.join(ch for ch in st if ch in singles) or singles[0] for st in strs]
        c2i = Char2idx(path)
        out[name + '_char2idx'] = [[st, [int(v) for v in c2i(st)]] for st in strs]
        if any(len(c) == 2 for c in chars):
            c2d = Char2idx(path, double_letter=True)
            out[name + '_char2idx_double'] = [[st, [int(v) for v in c2d(st)]] for st in strs]
    words = ['w%d' % i for i in range(12)]
    wl = []
    for _ in range(12):
        a = [words[i] for i in rng.randint(0, 12, size=rng.randint(1, 9))]
        b = [w for w in a if rng.rand() > 0.2] + [words[i] for i in rng.randint(0, 12, size=rng.randint(0, 3))]
        wl.append([a, b, float(compute_wer(ref=a, hyp=b, normalize=True)), int(compute_wer(ref=a, hyp=b, normalize=False))])
    out['compute_wer'] = wl
    with open(os.path.join(HERE, 'labels_v1.json'), 'w') as f:
        json.dump(out, f)
    print('labels_v1.json written')


def make_controller():
    """Learning-rate trajectories of the reference's utils/training/learning_rate_controller.py Controller on seeded
    score sequences (both directions, patience 0-3, different start epochs)."""
    import json
    sys.path.insert(0, REF)
    from utils.training.learning_rate_controller import Controller
    rng = np.random.RandomState(7)
    runs = []
    for lower_better in (True, False):
        for patient in (0, 1, 2, 3):
            for start in (1, 3, 6):
                vals = np.round(np.abs(np.cumsum(rng.randn(25) * 0.05) + (0.8 if lower_better else 0.3)), 4).tolist()
                c = Controller(learning_rate_init=1e-3, decay_start_epoch=start, decay_rate=0.5 + 0.1 * patient,
                               decay_patient_epoch=patient, lower_better=lower_better,
                               worst_value=1 if lower_better else 0)
                lr, traj = 1e-3, []
                for ep, v in enumerate(vals, 1):
                    lr = c.decay_lr(learning_rate=lr, epoch=ep, value=v)
                    traj.append(lr)
                runs.append(dict(lower_better=lower_better, patient=patient, start=start, rate=0.5 + 0.1 * patient,
                                 worst=1 if lower_better else 0, values=vals, lrs=traj))
    with open(os.path.join(HERE, 'controller_v1.json'), 'w') as f:
        json.dump(runs, f)
    print('controller_v1.json:', len(runs), 'runs')


def make_sparsetensor():
    """utils/io/labels/sparsetensor.py list2sparsetensor / sparsetensor2list (tensorflow stubbed: only
    tf.SparseTensorValue is touched, in an isinstance check) on seeded padded label batches."""
    import json
    sys.path.insert(0, REF)
    if 'tensorflow' not in sys.modules:
        tf = types.ModuleType('tensorflow')
        tf.SparseTensorValue = type('SparseTensorValue', (), {})
        sys.modules['tensorflow'] = tf
    elif not hasattr(sys.modules['tensorflow'], 'SparseTensorValue'):
        sys.modules['tensorflow'].SparseTensorValue = type('SparseTensorValue', (), {})
    from utils.io.labels.sparsetensor import list2sparsetensor, sparsetensor2list
    rng = np.random.RandomState(3)
    cases = []
    for B in (1, 2, 5, 16):
        for rep in range(3):
            lens = rng.randint(1, 9, size=B)
            dense = np.full((B, int(lens.max()) + rep), -1, dtype=np.int64)
            for b in range(B):
                dense[b, :lens[b]] = rng.randint(0, 40, size=lens[b])
            st = list2sparsetensor(dense, padded_value=-1)
            back = sparsetensor2list(st, B)
            cases.append(dict(dense=dense.tolist(), indices=st[0].tolist(), values=st[1].tolist(),
                              values_dtype=str(st[1].dtype), shape=st[2].tolist(),
                              back=[np.asarray(r).tolist() for r in back]))
    with open(os.path.join(HERE, 'sparsetensor_v1.json'), 'w') as f:
        json.dump(cases, f)
    print('sparsetensor_v1.json:', len(cases), 'cases')


class _OldNumpy(types.ModuleType):
    """numpy as the reference saw it (< 1.24): np.array() of a ragged list gives an object array instead of raising.
    Injected as `np` into the reference's dataset modules only."""

    def __init__(self):
        super(_OldNumpy, self).__init__('numpy_compat')

    def __getattr__(self, name):
        return getattr(np, name)

    def array(self, obj, *a, **k):
        try:
            return np.array(obj, *a, **k)
        except ValueError:
            seq = list(obj)
            out = np.empty(len(seq), dtype=object)
            for i, v in enumerate(seq):
                out[i] = v
            return out


def make_datasets():
    """Batches produced by the reference's OWN iterators (utils/dataset/{ctc,attention,joint_ctc_attention,
    multitask_ctc}.py DatasetBase.__next__) on a small generated corpus of .npy files, with `random` seeded, for
    every sampling mode (sorted window / shuffle / sequential), splice + stacking, 1 and 2 GPUs and the test-set
    (string label) path.  Stored: the corpus itself and, per configuration, every array of every batch."""
    import random
    import tempfile
    sys.path.insert(0, REF)
    import utils.io.inputs.frame_stacking as fs
    import utils.dataset.ctc as dctc
    import utils.dataset.attention as datt
    import utils.dataset.joint_ctc_attention as djoint
    import utils.dataset.multitask_ctc as dmulti
    shim = _OldNumpy()
    for m in (fs, dctc, datt, djoint, dmulti):
        m.np = shim
    rng = np.random.RandomState(42)
    tmp = tempfile.mkdtemp()
    N = 11
    feats = [rng.randn(rng.randint(4, 15), 6).astype(np.float32) for _ in range(N)]
    labs = [rng.randint(0, 9, size=rng.randint(1, 5)).astype(np.int32) for _ in range(N)]
    subs = [rng.randint(0, 4, size=rng.randint(1, 4)).astype(np.int32) for _ in range(N)]
    strs = [' '.join('p%d' % v for v in l) for l in labs]
    order = np.argsort([f.shape[0] for f in feats], kind='stable')        # subclasses list utterances by length
    paths = dict(inp=[], lab=[], sub=[], txt=[])
    for rank, i in enumerate(order):
        for key, arr in (('inp', feats[i]), ('lab', labs[i]), ('sub', subs[i]), ('txt', np.array(strs[i]))):
            pth = os.path.join(tmp, '%s_%02d.npy' % (key, rank))
            np.save(pth, arr)
            paths[key].append(pth)
    out = dict(corpus_order=order)
    for i in range(N):
        out['feat_%d' % i], out['lab_%d' % i], out['sub_%d' % i] = feats[i], labs[i], subs[i]
    out['strs'] = np.array(strs)
    eos_sos = {'<': 9, '>': 10}

    def make(cls, kind, cfg):
        class DS(cls):
            def __init__(self):
                super(DS, self).__init__()
                self.map_dict = dict(eos_sos)
                self.input_paths = np.array(paths['inp'])
                lab_key = 'txt' if cfg['is_test'] else 'lab'
                if kind == 'multitask':
                    self.label_main_paths = np.array(paths[lab_key])
                    self.label_sub_paths = np.array(paths['sub'])
                else:
                    self.label_paths = np.array(paths[lab_key])
                self.batch_size = cfg['batch_size'] * cfg['num_gpu']
                self.splice, self.num_stack, self.num_skip = cfg['splice'], cfg['num_stack'], cfg['num_skip']
                self.shuffle, self.sort_utt, self.sort_stop_epoch = cfg['shuffle'], cfg['sort_utt'], cfg['sort_stop']
                self.num_gpu, self.is_test, self.max_epoch = cfg['num_gpu'], cfg['is_test'], cfg['max_epoch']
                self.progressbar = False
                self.rest = set(range(N))
        return DS()

    cfgs = [
        dict(batch_size=3, num_gpu=1, splice=1, num_stack=1, num_skip=1, shuffle=False, sort_utt=True, sort_stop=1,
             is_test=False, max_epoch=3),
        dict(batch_size=4, num_gpu=1, splice=3, num_stack=2, num_skip=2, shuffle=True, sort_utt=False, sort_stop=None,
             is_test=False, max_epoch=2),
        dict(batch_size=2, num_gpu=2, splice=1, num_stack=3, num_skip=3, shuffle=False, sort_utt=False, sort_stop=None,
             is_test=False, max_epoch=2),
        dict(batch_size=1, num_gpu=1, splice=1, num_stack=1, num_skip=1, shuffle=False, sort_utt=False, sort_stop=None,
             is_test=True, max_epoch=1),
    ]
    kinds = [('ctc', dctc.DatasetBase), ('attention', datt.DatasetBase), ('joint', djoint.DatasetBase),
             ('multitask', dmulti.DatasetBase)]
    import json
    meta = []
    warnings.simplefilter('ignore')
    for kind, cls in kinds:
        for ci, cfg in enumerate(cfgs):
            random.seed(100 + ci)
            ds = make(cls, kind, cfg)
            nb = 0
            for data, is_new_epoch in ds:
                for fi, field in enumerate(data):
                    for gi in range(cfg['num_gpu']):
                        a = np.asarray(field[gi])
                        if a.dtype == object or a.dtype.kind in 'US':
                            a = np.array([str(v) for v in a.ravel()]).reshape(a.shape)
                        out['%s_c%d_b%d_f%d_g%d' % (kind, ci, nb, fi, gi)] = a
                out['%s_c%d_b%d_new' % (kind, ci, nb)] = np.array(bool(is_new_epoch))
                nb += 1
            meta.append(dict(kind=kind, cfg=ci, num_batches=nb, num_fields=len(data), epoch=int(ds.epoch)))
    out['meta'] = np.array(json.dumps(dict(cfgs=cfgs, runs=meta, sos=9, eos=10)))
    np.savez_compressed(os.path.join(HERE, 'datasets_v1.npz'), **out)
    print('datasets_v1.npz:', len(meta), 'runs,', sum(m['num_batches'] for m in meta), 'batches')


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs %s (build container only)' % REF)
    make_decoders()
    make_splice()
    make_labels()
    make_datasets()
    make_controller()
    make_sparsetensor()
