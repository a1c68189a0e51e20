"""CPU: host data formats on either side of the path pinned to
golden vectors produced by the reference's own functions (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_splice_and_stack_match_reference_golden():
    from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.splicing import do_splice
    from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.frame_stacking import stack_frame
    g = np.load(os.path.join(GOLD, 'splice_v1.npz'))
    for i in range(int(g['num_splice'])):
        splice, num_stack = g['s%d_cfg' % i]
        y = do_splice(g['s%d_in' % i], splice=int(splice), batch_size=2, num_stack=int(num_stack))
        assert y.shape == g['s%d_out' % i].shape
        assert np.array_equal(y, g['s%d_out' % i]), i
    for i in range(int(g['num_stack'])):
        num_stack, num_skip = g['f%d_cfg' % i]
        y = stack_frame(np.array([g['f%d_in' % i]]), int(num_stack), int(num_skip))
        assert np.array_equal(np.asarray(y[0], dtype=np.float64), g['f%d_out' % i]), i
    # the reference's in-file self test (splicing.py:76-88)
    assert do_splice(np.zeros((3, 100, 15)), splice=11, batch_size=3).shape == (3, 100, 165)


def test_dataset_iterator_contract():
    from tensorflow_end2end_speech_recognition_amd.utils.dataset.ctc import DatasetBase
    rng = np.random.RandomState(0)

    class DS(DatasetBase):
        def __init__(self, num_gpu):
            super(DS, self).__init__()
            self.input_paths = [rng.randn(rng.randint(5, 30), 6) for _ in range(23)]
            self.label_paths = [rng.randint(0, 5, size=rng.randint(1, 3)) for _ in range(23)]
            self.batch_size, self.splice, self.num_stack, self.num_skip = 4 * num_gpu, 3, 2, 2
            self.shuffle, self.sort_utt, self.sort_stop_epoch = False, True, 2
            self.num_gpu, self.is_test, self.max_epoch = num_gpu, False, 2
            self.rest = set(range(len(self.input_paths)))
    for num_gpu in (1, 2):
        ds = DS(num_gpu)
        n = 0
        for (inputs, labels, seq_len, names), new_epoch in ds:
            assert len(inputs) == num_gpu
            for gi in range(num_gpu):
                x, l, s = inputs[gi], labels[gi], seq_len[gi]
                assert x.dtype == np.float32 and x.shape[2] == 6 * 2 * 3
                assert x.shape[1] == inputs[0].shape[1]                      # global max T across shards
                for b in range(len(s)):
                    assert np.all(x[b, s[b]:] == 0)
                    assert s[b] >= (l[b] != -1).sum()                       # input length >= label length
            n += sum(len(s) for s in seq_len)
        assert n == 2 * 23 and ds.epoch == 2


def test_lr_controller():
    from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller
    c = Controller(1e-3, decay_start_epoch=2, decay_rate=0.5, decay_patient_epoch=1, lower_better=True)
    lr = 1e-3
    got = []
    for ep, v in enumerate([0.9, 0.8, 0.85, 0.86, 0.7, 0.71, 0.72], 1):
        lr = c.decay_lr(lr, ep, v)
        got.append(lr)
    assert got == [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 5e-4, 2.5e-4]


def test_lr_controller_matches_reference_trajectories():
    """24 learning-rate trajectories recorded from the reference's own Controller (controller_v1.json)."""
    import json
    from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller
    for run in json.load(open(os.path.join(GOLD, 'controller_v1.json'))):
        c = Controller(learning_rate_init=1e-3, decay_start_epoch=run['start'], decay_rate=run['rate'],
                       decay_patient_epoch=run['patient'], lower_better=run['lower_better'], worst_value=run['worst'])
        lr, got = 1e-3, []
        for ep, v in enumerate(run['values'], 1):
            lr = c.decay_lr(learning_rate=lr, epoch=ep, value=v)
            got.append(lr)
        assert got == run['lrs'], run


def test_sparsetensor_helpers_match_reference():
    """list2sparsetensor / sparsetensor2list against outputs of the reference's own functions
    (sparsetensor_v1.json), incl. the batch_size == 1 reshape and the value dtype."""
    import json
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor, sparsetensor2list
    for c in json.load(open(os.path.join(GOLD, 'sparsetensor_v1.json'))):
        dense = np.asarray(c['dense'], dtype=np.int64)
        st = list2sparsetensor(dense, padded_value=-1)
        assert st[0].tolist() == c['indices'] and st[1].tolist() == c['values'] and st[2].tolist() == c['shape']
        assert str(st[1].dtype) == c['values_dtype'] and st[0].dtype == np.int64 and st[2].dtype == np.int64
        back = sparsetensor2list(st, len(dense))
        assert [np.asarray(r).tolist() for r in back] == c['back']


def test_saver_roundtrip(tmp_path):
    """Saver.save / get_checkpoint_state / restore with the recipes' call shape (train_ctc.py:220-223,
    eval_ctc.py:74-86): TF variable names, `model.ckpt-<epoch>` prefixes, `checkpoint` index file."""
    import torch
    from tensorflow_end2end_speech_recognition_amd.utils.parameter import ParamStore
    from tensorflow_end2end_speech_recognition_amd.models.model_base import Optimizer
    from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, get_checkpoint_state

    class M(object):
        pass
    rng = np.random.RandomState(0)
    m = M()
    m.store = ParamStore(torch.device('cpu'))
    m.store.declare('blstm_hidden1/fw/lstm_cell/kernel', (7, 12), rng.randn(7, 12))
    m.store.declare('output/biases', (5,), rng.randn(5))
    m.store.finalize()
    m.optimizer = Optimizer('adam', 1e-3, m.store)
    m.optimizer.slot0.normal_()
    m.optimizer.global_step = 17
    assert get_checkpoint_state(str(tmp_path)) is None
    saver = Saver(max_to_keep=None)
    p1 = saver.save(m, os.path.join(str(tmp_path), 'model.ckpt'), global_step=3)
    assert p1.endswith('model.ckpt-3') and os.path.isfile(p1 + '.npz')
    want = {k: v.clone() for k, v in m.store.state_dict().items()}
    slot = m.optimizer.slot0.clone()
    m.store.flat.zero_()
    m.optimizer.slot0.zero_()
    m.optimizer.global_step = 0
    saver.save(m, os.path.join(str(tmp_path), 'model.ckpt'), global_step=4)
    ck = get_checkpoint_state(str(tmp_path))
    assert ck.model_checkpoint_path.endswith('model.ckpt-4') and len(ck.all_model_checkpoint_paths) == 2
    best = '/'.join(ck.model_checkpoint_path.split('/')[:-1]) + '/model.ckpt-' + str(3)     # eval_ctc.py:84-85
    saver.restore(m, best)
    for k, v in want.items():
        assert torch.equal(m.store[k], v)
    assert torch.equal(m.optimizer.slot0, slot) and m.optimizer.global_step == 17
    with np.load(best + '.npz') as z:
        assert z['blstm_hidden1/fw/lstm_cell/kernel'].shape == (7, 12)
    # resume into a model whose optimizer does not exist yet (the recipes create it lazily inside train()):
    # restore builds it from the checkpoint's record, so Adam's moments and bias-correction step continue
    from tensorflow_end2end_speech_recognition_amd.models.model_base import ModelBase
    fresh = ModelBase()
    fresh.store = ParamStore(torch.device('cpu'))
    fresh.store.declare('blstm_hidden1/fw/lstm_cell/kernel', (7, 12), np.zeros((7, 12)))
    fresh.store.declare('output/biases', (5,), np.zeros(5))
    fresh.store.finalize()
    assert fresh.optimizer is None
    saver.restore(fresh, best)
    assert fresh.optimizer is not None and fresh.optimizer.name == 'adam' and fresh.optimizer.global_step == 17
    assert torch.equal(fresh.optimizer.slot0, slot)
    other = ModelBase()
    other.store = fresh.store
    other.optimizer = Optimizer('rmsprop', 1e-3, other.store)
    with pytest.warns(UserWarning):
        saver.restore(other, best)
    with pytest.raises(ValueError):
        saver.restore(m, os.path.join(str(tmp_path), 'model.ckpt-9'))


def _golden_labels():
    import json
    return json.load(open(os.path.join(GOLD, 'labels_v1.json')))


def _write_map(path, table):
    with open(path, 'w') as f:
        for tok, idx in table:
            f.write('%s  %d\n' % (tok, idx))


def test_label_maps_match_reference_golden(tmp_path):
    """Phone2idx / Idx2phone / Char2idx / Idx2char (utils/io/labels/*.py) against outputs of the reference's own
    classes on its mapping files (tests/golden/labels_v1.json)."""
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.phone import Phone2idx, Idx2phone
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.character import Char2idx, Idx2char
    g = _golden_labels()
    for lt in ('phone61', 'phone48', 'phone39'):
        path = str(tmp_path / (lt + '.txt'))
        _write_map(path, [(p, i) for i, p in enumerate(g[lt + '_table'])])
        i2p, p2i = Idx2phone(path), Phone2idx(path)
        for seq, want in g[lt + '_idx2phone']:
            assert i2p(np.array(seq)) == want
        for phones, want in zip(*g[lt + '_phone2idx']):
            assert p2i(list(phones)).tolist() == want
    for name, kw in (('character', {}), ('character_capital_divide', dict(capital_divide=True, space_mark='_'))):
        path = str(tmp_path / (name + '.txt'))
        _write_map(path, g[name + '_table'])
        i2c = Idx2char(path, **kw)
        for seq, want in g[name + '_idx2char']:
            assert i2c(np.array(seq)) == want
        c2i = Char2idx(path)
        for st, want in g[name + '_char2idx']:
            assert [int(v) for v in c2i(st)] == want
        if (name + '_char2idx_double') in g:
            c2d = Char2idx(path, double_letter=True)
            for st, want in g[name + '_char2idx_double']:
                assert [int(v) for v in c2d(st)] == want


def test_phone_folding_and_error_rates_match_reference_golden():
    """Map2phone39's built-in Lee & Hon table phone by phone against the reference's class on its
    phone2phone.txt; compute_wer against the reference's numpy implementation; PER / CER / wer_align
    consistency with the unit-cost edit distance."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples.timit.metrics.mapping import Map2phone39
    from tensorflow_end2end_speech_recognition_amd.utils.evaluation import edit_distance as ed
    g = _golden_labels()
    for lt in ('phone61', 'phone48', 'phone39'):
        m = Map2phone39(lt)
        for ph, want in g[lt + '_to39_table']:
            assert m([ph]) == want, (lt, ph)
        for seq, want in g[lt + '_to39_seq']:
            assert m(list(seq)) == want
    for ref, hyp, wer, dist in g['compute_wer']:
        assert abs(ed.compute_wer(ref=ref, hyp=hyp, normalize=True) - wer) < 1e-12
        assert int(ed.compute_wer(ref=ref, hyp=hyp, normalize=False)) == dist == ed.levenshtein(ref, hyp)
        s, i, d = ed.wer_align(ref, hyp)
        assert s + i + d == dist and len(ref) - d + i == len(hyp)
    assert ed.compute_per(['a', 'b', 'c', 'd'], ['a', 'x', 'd'], normalize=True) == 2 / 4
    assert ed.compute_cer('kitten', 'sitting', normalize=False) == 3
    assert abs(ed.compute_cer('kitten', 'sitting') - 3 / 7) < 1e-12


class _FakeDataset(object):
    """Two batches of two utterances; yields (data, is_new_epoch) like the reference's DatasetBase."""

    def __init__(self, batches, label_type, padded_value=-1):
        self.batches, self.label_type, self.padded_value, self.batch_size = batches, label_type, padded_value, 2

    def reset(self):
        pass

    def __len__(self):
        return sum(b[0][0].shape[0] for b in self.batches)

    def __iter__(self):
        for i, b in enumerate(self.batches):
            yield b, i == len(self.batches) - 1


class _FakeModel(object):
    """Stands in for models.ctc.CTC: the 'logits' are the label rows to return, decoding is the identity."""

    def compute_loss(self, inputs, labels, seq_len, keep_prob=1.0, is_training=False):
        return None, inputs

    def decoder(self, logits, seq_len, beam_width=1):
        from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
        return list2sparsetensor(np.asarray(logits)[:, :, 0].astype(np.int64), padded_value=-1)


def test_eval_loops_on_fake_model(tmp_path):
    """do_eval_per / do_eval_cer (examples/timit/metrics/ctc.py) end to end on CPU with a stand-in model: the
    index->string maps, 61->39 folding, punctuation stripping and the PER/CER/WER means."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples.timit.metrics.ctc import do_eval_per, do_eval_cer
    from tensorflow_end2end_speech_recognition_amd.utils.evaluation import edit_distance as ed
    g = _golden_labels()
    for lt in ('phone61', 'phone39'):
        _write_map(str(tmp_path / (lt + '.txt')), [(p, i) for i, p in enumerate(g[lt + '_table'])])
    _write_map(str(tmp_path / 'character.txt'), g['character_table'])
    t61 = g['phone61_table']

    def rows(*seqs):
        L = max(len(q) for q in seqs)
        return np.array([list(q) + [-1] * (L - len(q)) for q in seqs], dtype=np.int32)

    def batch(hyp, true):
        return ([hyp[:, :, None].astype(np.float32)], [true], [np.full(len(hyp), hyp.shape[1], np.int32)], [None])

    ix = dict((p, i) for i, p in enumerate(t61))
    # utterance 0: identical; 1: 'ao' vs 'aa' fold to the same phone -> 0; 2: one substitution over 3; 3: closure dropped by folding
    hyp = [[ix['sil'] if 'sil' in ix else ix['h#'], ix['aa'], ix['b']], [ix['ao'], ix['d']], [ix['iy'], ix['k'], ix['s']],
           [ix['aa'], ix['q'], ix['t']]]
    tru = [hyp[0], [ix['aa'], ix['d']], [ix['iy'], ix['g'], ix['s']], [ix['aa'], ix['t']]]
    ds = _FakeDataset([batch(rows(*hyp[:2]), rows(*tru[:2])), batch(rows(*hyp[2:]), rows(*tru[2:]))], 'phone61')
    per = do_eval_per(None, None, None, _FakeModel(), ds, 'phone61', map_dir=str(tmp_path))
    assert abs(per - (0 + 0 + 1 / 3 + 0) / 4) < 1e-12
    cmap = dict(g['character_table'])
    enc = lambda st: [cmap[c] for c in st]
    hyp_s, tru_s = ['the_cat', 'a__dog', "it's_fine", 'one_two'], ['the_cat', 'a_dig', 'its_fine', 'one_too_x']
    ds = _FakeDataset([batch(rows(*map(enc, hyp_s[:2])), rows(*map(enc, tru_s[:2]))),
                       batch(rows(*map(enc, hyp_s[2:])), rows(*map(enc, tru_s[2:])))], 'character')
    cer, wer = do_eval_cer(None, None, _FakeModel(), ds, 'character', map_dir=str(tmp_path))
    want_cer = (0 + 1 / 4 + 0 + ed.levenshtein('onetwo', 'onetoox') / 7) / 4
    want_wer = (0 + 1 / 2 + 0 + 2 / 3) / 4
    assert abs(cer - want_cer) < 1e-12 and abs(wer - want_wer) < 1e-12


def test_dataset_device_assembly_yields_raw_features():
    """device_assembly=True: the iterator hands over the raw zero-padded features and raw frame counts (the
    stacking / splicing then runs in utils/io/inputs/device.py assemble(), GPU test test_device_batch_assembly);
    sampling order and labels are the host path's."""
    from tensorflow_end2end_speech_recognition_amd.utils.dataset.ctc import DatasetBase
    feats = [np.random.RandomState(i).randn(5 + 3 * i, 6) for i in range(7)]
    labs = [np.arange(1 + i % 3) for i in range(7)]

    class DS(DatasetBase):
        def __init__(self, device_assembly):
            super(DS, self).__init__()
            self.input_paths, self.label_paths = feats, labs
            self.batch_size, self.splice, self.num_stack, self.num_skip = 3, 3, 2, 2
            self.shuffle, self.sort_utt, self.sort_stop_epoch = False, False, None
            self.num_gpu, self.is_test, self.max_epoch = 1, False, 1
            self.rest = set(range(len(feats)))
            self.device_assembly = device_assembly
    host, dev = DS(False), DS(True)
    for ((xh, lh, sh, nh), _), ((xd, ld, sd, nd), _) in zip(host, dev):
        assert np.array_equal(lh, ld) and list(nh[0]) == list(nd[0])
        ids = [int(n) for n in nd[0]]
        assert xd.shape[-1] == 6 and xh.shape[-1] == 6 * 2 * 3
        for r, i in enumerate(ids):
            T = feats[i].shape[0]
            assert sd[0][r] == T and sh[0][r] == -(-T // 2)
            assert np.array_equal(xd[0][r, :T], feats[i].astype(np.float32)) and not xd[0][r, T:].any()


def test_generated_mapping_files_match_reference_tables(tmp_path):
    """examples/timit/metrics/mapping_files.py writes the token tables instead of shipping the reference's files;
    every generated table equals the one read from the reference's own mapping file (labels_v1.json), and
    Map2phone39 on the generated phone2phone.txt folds like the reference's class on its file."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples.timit.metrics.mapping_files import write_mapping_files
    from examples.timit.metrics.mapping import Map2phone39
    g = _golden_labels()
    d = write_mapping_files(str(tmp_path / 'maps'))
    for lt in ('phone61', 'phone48', 'phone39'):
        rows = [l.split() for l in open(os.path.join(d, lt + '.txt'))]
        assert [r[0] for r in rows] == g[lt + '_file_tokens'] == g[lt + '_table'] + ['<', '>']
        assert [int(r[1]) for r in rows] == list(range(len(rows)))
    for name in ('character', 'character_capital_divide'):
        rows = [l.split() for l in open(os.path.join(d, name + '.txt'))]
        assert [[r[0], int(r[1])] for r in rows] == g[name + '_table']
    for lt in ('phone61', 'phone48'):
        m = Map2phone39(lt, os.path.join(d, 'phone2phone.txt'))
        for ph, want in g[lt + '_to39_table']:
            assert m([ph]) == want


@pytest.mark.parametrize('kind', ['ctc', 'attention', 'joint', 'multitask'])
def test_dataset_iterators_match_reference_batches(tmp_path, kind):
    """utils/dataset/{ctc,attention,joint_ctc_attention,multitask_ctc}.py: every array of every batch equals what
    the reference's own DatasetBase.__next__ produced on the same corpus with the same `random` seed
    (tests/golden/datasets_v1.npz: sorted-window / shuffle / sequential sampling, splice + stacking, 2-GPU split,
    string labels of the test set)."""
    import importlib
    import json
    import random
    z = np.load(os.path.join(GOLD, 'datasets_v1.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    order = z['corpus_order']
    N = len(order)
    strs = z['strs']
    paths = dict(inp=[], lab=[], sub=[], txt=[])
    for rank, i in enumerate(order):
        for key, arr in (('inp', z['feat_%d' % i]), ('lab', z['lab_%d' % i]), ('sub', z['sub_%d' % i]),
                         ('txt', np.array(str(strs[i])))):
            pth = str(tmp_path / ('%s_%02d.npy' % (key, rank)))
            np.save(pth, arr)
            paths[key].append(pth)
    mod = importlib.import_module('tensorflow_end2end_speech_recognition_amd.utils.dataset.' +
                                  dict(ctc='ctc', attention='attention', joint='joint_ctc_attention',
                                       multitask='multitask_ctc')[kind])

    def make(cfg):
        class DS(mod.DatasetBase):
            def __init__(self):
                super(DS, self).__init__()
                self.map_dict = {'<': meta['sos'], '>': meta['eos']}
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
                self.rest = set(range(N))
        return DS()

    runs = [r for r in meta['runs'] if r['kind'] == kind]
    assert len(runs) == len(meta['cfgs'])
    for run in runs:
        cfg = meta['cfgs'][run['cfg']]
        random.seed(100 + run['cfg'])
        ds = make(cfg)
        nb = 0
        for data, is_new_epoch in ds:
            assert len(data) == run['num_fields']
            assert bool(z['%s_c%d_b%d_new' % (kind, run['cfg'], nb)]) == bool(is_new_epoch)
            for fi, field in enumerate(data):
                assert len(field) == cfg['num_gpu']
                for gi in range(cfg['num_gpu']):
                    want = z['%s_c%d_b%d_f%d_g%d' % (kind, run['cfg'], nb, fi, gi)]
                    got = np.asarray(field[gi])
                    if want.dtype.kind in 'US':
                        got = np.array([str(v) for v in got.ravel()]).reshape(got.shape)
                    assert got.shape == want.shape, (run, nb, fi, got.shape, want.shape)
                    assert np.array_equal(got, want), (run, nb, fi)
            nb += 1
        assert nb == run['num_batches'] and ds.epoch == run['epoch']


def test_small_host_utilities(tmp_path, capsys):
    """utils/directory.py, measure_time_func.py, progressbar.py, training/plot.py and compute_edit_distance: the
    reference's helper calls keep working."""
    from tensorflow_end2end_speech_recognition_amd.utils.directory import mkdir, mkdir_join
    from tensorflow_end2end_speech_recognition_amd.utils.measure_time_func import measure_time
    from tensorflow_end2end_speech_recognition_amd.utils.progressbar import wrap_iterator, wrap_generator
    from tensorflow_end2end_speech_recognition_amd.utils.training.plot import plot_loss, plot_ler
    from tensorflow_end2end_speech_recognition_amd.utils.evaluation.edit_distance import compute_edit_distance
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor
    root = str(tmp_path / 'runs')
    assert mkdir(root) == root and os.path.isdir(root) and mkdir(None) is None and mkdir_join(None, 'a') is None
    p = mkdir_join(root, 'ctc', 'phone61', 'model.ckpt')
    assert p == os.path.join(root, 'ctc', 'phone61', 'model.ckpt') and os.path.isdir(os.path.dirname(p))
    assert not os.path.exists(p)                                    # a dotted component is a file name

    @measure_time
    def f(a, b=1):
        return a + b
    assert f(2, b=3) == 5 and 'Takes' in capsys.readouterr().out
    assert list(wrap_iterator(range(3), False)) == [0, 1, 2] and list(wrap_generator(iter([1, 2]), False, 2)) == [1, 2]
    plot_loss([3.0, 2.0], [3.5, 2.5], [10, 20], root)
    plot_ler([0.9, 0.5], [0.95, 0.6], [10, 20], 'phone61', root)
    assert open(os.path.join(root, 'loss.csv')).read().splitlines()[1] == '10,3.000000,3.500000'
    assert open(os.path.join(root, 'ler.csv')).read().splitlines()[2] == '20,0.500000,0.600000'
    true = list2sparsetensor(np.array([[1, 2, 3, -1], [4, 4, 5, 6]]), -1)
    pred = list2sparsetensor(np.array([[1, 3, -1, -1], [4, 4, 5, 6]]), -1)
    # the reference's function swaps its arguments before tf.edit_distance: divided by the PREDICTION's length
    assert np.allclose(compute_edit_distance(None, true, pred), [1 / 2, 0.0])


def test_edit_distance_matches_tensorflow_documented_cases():
    """tf_edit_distance (normalised, per utterance) on the cases of tf.edit_distance's own documentation:
    truth [b, c] vs hypothesis [b] -> 0.5 (one addition), truth [a] vs no hypothesis -> 1.0, no truth -> inf."""
    from tensorflow_end2end_speech_recognition_amd.utils.evaluation.edit_distance import tf_edit_distance
    # rows: 0 = no truth / hyp [7]; 1 = truth [1, 2] / hyp [1]; 2 = truth [0] / no hypothesis
    truth = [np.array([[1, 0], [1, 1], [2, 0]], dtype=np.int64), np.array([1, 2, 0], dtype=np.int32),
             np.array([3, 2], dtype=np.int64)]
    hyp = [np.array([[0, 0], [1, 0]], dtype=np.int64), np.array([7, 1], dtype=np.int32), np.array([3, 1], dtype=np.int64)]
    d = tf_edit_distance(hyp, truth)
    assert np.isinf(d[0]) and abs(d[1] - 0.5) < 1e-12 and abs(d[2] - 1.0) < 1e-12
    assert tf_edit_distance(hyp, truth, normalize=False).tolist() == [1.0, 1.0, 1.0]
