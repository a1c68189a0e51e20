#!/usr/bin/env python
"""Evaluate a trained attention (or joint CTC-attention) model on the TIMIT test set -- the recipe of
examples/timit/evaluation/eval_attention.py.

    python examples/timit/evaluation/eval_attention.py <model_path> [--epoch E] [--joint]

<model_path> is a run directory of train_attention.py / train_joint_ctc_attention.py."""
import argparse
import sys
from os.path import abspath, dirname, isfile, join

import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.timit.metrics.attention import do_eval_per, do_eval_cer                                        # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                         # noqa: E402
from examples.timit.training._common import NUM_CLASSES                                                      # noqa: E402
from examples.timit.training.train_attention import make_datasets, model_kwargs                              # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, get_checkpoint_state  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('model_path')
    ap.add_argument('--epoch', type=int, default=-1, help='the epoch to restore')
    ap.add_argument('--eval_batch_size', type=int, default=1)
    ap.add_argument('--joint', action='store_true', help='the run is a joint CTC-attention model')
    ap.add_argument('--device', default=None)
    args = ap.parse_args(argv)
    with open(join(args.model_path, 'config.yml'), 'r') as f:
        params = yaml.safe_load(f)['param']
    if args.device:
        params['device'] = args.device
    params['num_classes'] = NUM_CLASSES[params['label_type']]
    if args.joint:
        from examples.timit.data.load_dataset_joint_ctc_attention import Dataset
        from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention
        model = JointCTCAttention(lambda_weight=params['lambda_weight'], **model_kwargs(params))
    else:
        from examples.timit.data.load_dataset_attention import Dataset
        from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq
        model = AttentionSeq2Seq(**model_kwargs(params))
    ckpt = get_checkpoint_state(args.model_path)
    if ckpt is None:
        raise ValueError('There are not any checkpoints.')
    path = ckpt.model_checkpoint_path if args.epoch == -1 else join(args.model_path, 'model.ckpt-' + str(args.epoch))
    Saver().restore(model, path)
    map_dir = params.get('map_dir') or join(args.model_path, 'mapping_files')
    if not isfile(join(map_dir, 'phone2phone.txt')):
        write_mapping_files(map_dir)
    _, _, test_data = make_datasets(Dataset, params, map_dir)
    print('Test Data Evaluation:')
    ev = dict(session=None, decode_op=None, model=model, dataset=test_data, label_type=params['label_type'],
              is_test=True, eval_batch_size=args.eval_batch_size, map_dir=map_dir, is_jointctcatt=args.joint)
    if 'char' in params['label_type']:
        cer, wer = do_eval_cer(**ev)
        print('  CER: %f %%' % (cer * 100))
        print('  WER: %f %%' % (wer * 100))
        return cer
    per = do_eval_per(per_op=None, **ev)
    print('  PER: %f %%' % (per * 100))
    return per


if __name__ == '__main__':
    main()
