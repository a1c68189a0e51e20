#!/usr/bin/env python
"""End-to-end CTC recipe on synthetic utterances, with the call sequence of the reference's
examples/timit/training/train_ctc.py:38-300 (dataset iterator -> compute_loss -> train ->
decoder / compute_ler every print_step -> learning-rate controller -> Saver on a new best) on the
MI355X path.  The corpus is generated: each label owns a fixed random feature vector and is held for
a few frames with noise on top, so a small BLSTM learns it in a few epochs.

    python examples/synthetic/train_ctc.py [--epochs 6] [--save_path /tmp/ctc_synth]
"""
import argparse
import os
import sys
from os.path import join

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC                           # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.dataset.ctc import DatasetBase                # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, get_checkpoint_state  # noqa: E402


class SyntheticDataset(DatasetBase):
    def __init__(self, num_utt, num_classes, feat_dim, batch_size, seed, max_epoch=None, shuffle=True,
                 sort_utt=False, is_test=False, num_gpu=1, protos=None):
        super(SyntheticDataset, self).__init__()
        rng = np.random.RandomState(seed)
        self.protos = protos if protos is not None else rng.randn(num_classes, feat_dim).astype(np.float32)
        self.input_paths, self.label_paths = [], []
        for _ in range(num_utt):
            lab = rng.randint(0, num_classes, size=rng.randint(3, 9))
            frames = []
            for c in lab:
                frames.append(np.repeat(self.protos[c][None], rng.randint(2, 5), axis=0))
                frames.append(np.zeros((rng.randint(0, 2), feat_dim), np.float32))
            x = np.concatenate(frames, 0)
            self.input_paths.append((x + 0.3 * rng.randn(*x.shape)).astype(np.float32))
            self.label_paths.append(lab.astype(np.int32))
        self.batch_size, self.splice, self.num_stack, self.num_skip = batch_size * num_gpu, 1, 1, 1
        self.shuffle, self.sort_utt, self.sort_stop_epoch = shuffle, sort_utt, None
        self.num_gpu, self.is_test, self.max_epoch = num_gpu, is_test, max_epoch
        self.rest = set(range(num_utt))


def evaluate(model, dataset):
    """examples/timit/metrics/ctc.py:20-124 in small: mean label error rate over a pass of `dataset`."""
    tot, n = 0.0, 0
    dataset.reset()
    epoch0 = dataset.epoch
    for (inputs, labels, seq_len, _), new_epoch in dataset:
        loss, logits = model.compute_loss(inputs[0], list2sparsetensor(labels[0], padded_value=-1), seq_len[0],
                                          keep_prob=1.0, is_training=False)
        dec = model.decoder(logits, seq_len[0], beam_width=1)
        tot += model.compute_ler(dec, list2sparsetensor(labels[0], padded_value=-1)) * len(seq_len[0])
        n += len(seq_len[0])
        if new_epoch or dataset.epoch > epoch0:
            break
    return tot / max(n, 1)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=6)
    ap.add_argument('--save_path', default='/tmp/ctc_synth')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--units', type=int, default=256)
    args = ap.parse_args(argv)
    params = dict(num_classes=12, input_size=24, num_units=args.units, num_layers=2, batch_size=32, optimizer='adam',
                  learning_rate=2e-3, dropout=0.1, clip_grad_norm=5.0, clip_activation=50, print_step=20,
                  decay_start_epoch=3, decay_rate=0.7, decay_patient_epoch=1)
    train = SyntheticDataset(512, params['num_classes'], params['input_size'], params['batch_size'], seed=1,
                             max_epoch=args.epochs, sort_utt=True)
    train.sort_stop_epoch = 2
    dev = SyntheticDataset(64, params['num_classes'], params['input_size'], params['batch_size'], seed=2,
                           shuffle=False, protos=train.protos)
    model = CTC(encoder_type='blstm', input_size=params['input_size'], num_units=params['num_units'],
                num_layers=params['num_layers'], num_classes=params['num_classes'],
                parameter_init=0.1, clip_grad_norm=params['clip_grad_norm'],
                clip_activation=params['clip_activation'], dtype=args.dtype)
    model.save_path = args.save_path
    saver = Saver(max_to_keep=None)
    lr_controller = Controller(params['learning_rate'], params['decay_start_epoch'], params['decay_rate'],
                               params['decay_patient_epoch'], lower_better=True)
    lr = params['learning_rate']
    best = 1.0
    history = []
    for step, ((inputs, labels, seq_len, _), is_new_epoch) in enumerate(train):
        loss, logits = model.compute_loss(inputs[0], list2sparsetensor(labels[0], padded_value=-1), seq_len[0],
                                          keep_prob=1.0 - params['dropout'])
        model.train(loss, optimizer=params['optimizer'], learning_rate=lr)
        if (step + 1) % params['print_step'] == 0:
            print('Step %d (epoch %.3f): loss = %.3f / lr = %.5f' % (step + 1, train.epoch_detail, loss.item(), lr))
        if is_new_epoch:
            ler = evaluate(model, dev)
            history.append(ler)
            print('=== epoch %d: dev LER %.4f ===' % (train.epoch, ler))
            if ler < best:
                best = ler
                print('Model saved in file: %s' % saver.save(model, join(model.save_path, 'model.ckpt'),
                                                           global_step=train.epoch))
            lr = lr_controller.decay_lr(lr, train.epoch, ler)
    ckpt = get_checkpoint_state(model.save_path)
    return dict(history=history, best=best, checkpoint=ckpt.model_checkpoint_path if ckpt else None, model=model,
                dev=dev)


if __name__ == '__main__':
    out = main()
    print('dev LER per epoch:', ['%.3f' % v for v in out['history']], 'best checkpoint:', out['checkpoint'])
