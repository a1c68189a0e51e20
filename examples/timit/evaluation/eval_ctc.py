#!/usr/bin/env python
"""Evaluate a trained CTC model on the TIMIT test set -- the recipe of examples/timit/evaluation/eval_ctc.py:30-164.

    python examples/timit/evaluation/eval_ctc.py <model_path> [--epoch E] [--beam_width W] [--eval_batch_size B]

<model_path> is a run directory of train_ctc.py (config.yml + checkpoint index + model.ckpt-<epoch>.npz).  The model
is rebuilt from config.yml, the checkpoint of `--epoch` (-1: the latest) is restored, and PER on 39 phones or
CER/WER of the test set is printed."""
import argparse
import sys
from os.path import abspath, dirname, isfile, join

import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.timit.data.load_dataset_ctc import Dataset                                           # noqa: E402
from examples.timit.metrics.ctc import do_eval_per, do_eval_cer                                    # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                               # noqa: E402
from examples.timit.training.train_ctc import build_model                                          # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, get_checkpoint_state  # noqa: E402


def do_eval(model, params, epoch, eval_batch_size, beam_width, model_path):
    """:30-108."""
    is_char = 'char' in params['label_type']
    test_data = Dataset(data_type='test', label_type=params['label_type'] if is_char else 'phone39', batch_size=1,
                        splice=params['splice'], num_stack=params['num_stack'], num_skip=params['num_skip'],
                        shuffle=False, dataset_root=params.get('dataset_root'))
    ckpt = get_checkpoint_state(model_path)
    if ckpt is None:
        raise ValueError('There are not any checkpoints.')           # :75-76
    path = ckpt.model_checkpoint_path
    if epoch != -1:
        path = join(model_path, 'model.ckpt-' + str(epoch))
    Saver().restore(model, path)
    map_dir = params.get('map_dir') or join(model_path, 'mapping_files')
    if not isfile(join(map_dir, 'phone2phone.txt')):
        write_mapping_files(map_dir)
    print('Test Data Evaluation:')
    ev = dict(model=model, dataset=test_data, label_type=params['label_type'], is_test=True,
              eval_batch_size=eval_batch_size, map_dir=map_dir, beam_width=beam_width)
    if is_char:
        cer, wer = do_eval_cer(session=None, decode_op=None, **ev)
        print('  CER: %f %%' % (cer * 100))
        print('  WER: %f %%' % (wer * 100))
        return cer, wer
    per = do_eval_per(session=None, decode_op=None, per_op=None, **ev)
    print('  PER: %f %%' % (per * 100))
    return per


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('model_path')
    ap.add_argument('--epoch', type=int, default=-1, help='the epoch to restore')
    ap.add_argument('--beam_width', type=int, default=20)
    ap.add_argument('--eval_batch_size', type=int, default=1)
    ap.add_argument('--device', default=None)
    args = ap.parse_args(argv)
    with open(join(args.model_path, 'config.yml'), 'r') as f:
        params = yaml.safe_load(f)['param']
    if args.device:
        params['device'] = args.device
    model = build_model(params)
    model.save_path = args.model_path
    return do_eval(model, params, args.epoch, args.eval_batch_size, args.beam_width, args.model_path)


if __name__ == '__main__':
    main()
