#!/usr/bin/env python
"""Train the hierarchical (multitask) CTC model on Librispeech, data parallel -- the recipe of
examples/librispeech/training/train_multitask_ctc.py: word targets on the top layer, character targets on layer
`num_layers_sub`, one process per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/librispeech/training/train_multitask_ctc.py <config.yml> <model_save_path>

Tower arithmetic as train_ctc.py (multi_gpu.tower_step_with); rank 0 scores the dev set (WER of the word head, CER of
the character head, the latter the early-stopping metric as in the reference, :300-360), checkpoints on a new best."""
import os
import random
import sys
import time
from os.path import abspath, dirname, isfile, join

import torch
import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.librispeech.data.load_dataset_multitask_ctc import Dataset                                     # noqa: E402
from examples.librispeech.metrics.ctc import do_eval_cer, do_eval_wer                                        # noqa: E402
from examples.librispeech.training.train_ctc import _bcast                                                   # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                         # noqa: E402
from examples.timit.training._common import new_run_directory                                                # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.multitask_ctc import MultitaskCTC                  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor         # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu                               # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, sync_point                        # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller     # noqa: E402

NUM_WORDS = {'train100h': 7213, 'train460h': 18641, 'train960h': 26642}          # word_freq10 (:443-449)
NUM_CHARS = {'character': {'train100h': 28, 'train460h': 28, 'train960h': 28},
             'character_capital_divide': {'train100h': 72, 'train460h': 77, 'train960h': 77}}


def do_train(model, params, rank, world):
    map_dir = params.get('map_dir') or join(model.save_path, 'mapping_files')
    if rank == 0 and not isfile(join(map_dir, 'character.txt')):
        write_mapping_files(map_dir)
    kw = dict(train_data_size=params['train_data_size'], label_type_main=params['label_type_main'],
              label_type_sub=params['label_type_sub'], batch_size=params['batch_size'], splice=params['splice'],
              num_stack=params['num_stack'], num_skip=params['num_skip'], dataset_root=params.get('dataset_root'))
    seed = params.get('seed', 0)
    train_data = Dataset(data_type='train', max_epoch=params['num_epoch'], sort_utt=True,
                         sort_stop_epoch=params['sort_stop_epoch'], num_gpu=world, **kw)
    train_data.rng = random.Random(seed)
    small = params['train_data_size'] in ['train100h', 'train460h']
    if rank == 0:
        dev_data = Dataset(data_type='dev_clean' if small else 'dev_other', shuffle=False, **kw)
        dev_data.rng = random.Random(seed + 1)
    multi_gpu.broadcast_parameters(model.store)
    optimizer = model._set_optimizer(params['optimizer'], params['learning_rate'])
    model.optimizer = optimizer
    lr_controller = Controller(learning_rate_init=params['learning_rate'],
                               decay_start_epoch=params['decay_start_epoch'], decay_rate=params['decay_rate'],
                               decay_patient_epoch=params['decay_patient_epoch'], lower_better=True)
    saver = Saver(max_to_keep=None)
    keep_prob = 1 - float(params['dropout'])
    learning_rate = float(params['learning_rate'])
    print_step = max(1, int(params['print_step'] / world))
    best, not_improved = 1, 0
    result = dict(metric_dev=[], checkpoints=[], steps=0, losses=[])
    start_step = time.time()
    for step, (data, is_new_epoch) in enumerate(train_data):
        inputs, labels_main, labels_sub, inputs_seq_len, _ = data
        x, ym, ys, sl = inputs[rank], labels_main[rank], labels_sub[rank], inputs_seq_len[rank]

        def loss_fn():
            if len(x) == 0:
                return None
            return model.compute_loss(x, list2sparsetensor(ym, padded_value=-1), list2sparsetensor(ys, padded_value=-1),
                                      sl, keep_prob)[0]
        loss = multi_gpu.tower_step_with(model, optimizer, loss_fn, learning_rate=learning_rate)
        result['steps'] = step + 1
        result['losses'].append(float(loss))
        if (step + 1) % print_step == 0 and rank == 0:
            print('Step %d (epoch: %.3f): loss = %.3f / lr = %.5f (%.3f min)' %
                  (step + 1, train_data.epoch_detail, float(loss), learning_rate, (time.time() - start_step) / 60))
            sys.stdout.flush()
            start_step = time.time()
        if is_new_epoch:
            sync_point()          # pending asynchronous error checks of this epoch's steps are raised here
            stop = False
            if rank == 0:
                print('-----EPOCH:%d-----' % train_data.epoch)
            if train_data.epoch >= params['eval_start_epoch']:
                if rank == 0:
                    print('=== Dev Data Evaluation ===')
                    ev = dict(session=None, decode_ops=None, model=model, dataset=dev_data, eval_batch_size=1,
                              map_dir=map_dir, beam_width=params['beam_width'], is_multitask=True)
                    cer, _ = do_eval_cer(label_type=params['label_type_sub'], **ev)
                    print('  CER (sub): %f %%' % (cer * 100))
                    if isfile(join(map_dir, 'word_' + params['train_data_size'] + '.txt')):
                        wer = do_eval_wer(train_data_size=params['train_data_size'], **ev)
                        print('  WER (main): %f %%' % (wer * 100))
                    result['metric_dev'].append(cer)
                    if cer < best:
                        best, not_improved = cer, 0
                        print('■■■ ↑Best Score (CER)↑ ■■■')
                        path = saver.save(model, join(model.save_path, 'model.ckpt'), global_step=train_data.epoch)
                        result['checkpoints'].append(path)
                        print('Model saved in file: %s' % path)
                    else:
                        not_improved += 1
                    stop = not_improved == params['not_improved_patient_epoch']
                    if not stop:
                        learning_rate = lr_controller.decay_lr(learning_rate=learning_rate, epoch=train_data.epoch,
                                                               value=cer)
                stop, learning_rate = _bcast((stop, learning_rate))
            if stop:
                break
    if rank == 0:
        sync_point()
        with open(join(model.save_path, 'complete.txt'), 'w') as f:
            f.write('')
    return result


def main(config_path, model_save_path, log_to_file=True):
    with open(config_path, 'r') as f:
        params = yaml.safe_load(f)['param']
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    device = params.get('device') or 'cuda:%d' % local_rank
    if device.startswith('cuda'):
        torch.cuda.set_device(torch.device(device))
    rank, world = multi_gpu.init_process_group(device)
    params.setdefault('num_classes_main', NUM_WORDS[params['train_data_size']])
    params.setdefault('num_classes_sub', NUM_CHARS[params['label_type_sub']][params['train_data_size']])
    model = MultitaskCTC(encoder_type=params['encoder_type'], input_size=params['input_size'],
                         num_units=params['num_units'], num_layers_main=params['num_layers_main'],
                         num_layers_sub=params['num_layers_sub'], num_classes_main=params['num_classes_main'],
                         num_classes_sub=params['num_classes_sub'], main_task_weight=params['main_task_weight'],
                         lstm_impl=params['lstm_impl'], use_peephole=params['use_peephole'], splice=params['splice'],
                         parameter_init=params['weight_init'], clip_grad_norm=params['clip_grad_norm'],
                         clip_activation=params['clip_activation'], num_proj=params['num_proj'],
                         weight_decay=params['weight_decay'], dtype=params.get('dtype', 'bf16'), device=device,
                         seed=params.get('seed', 0))
    model.name += '_%d_main%d_sub%d_%s_lr%s_main%s' % (params['num_units'], params['num_layers_main'],
                                                      params['num_layers_sub'], params['optimizer'],
                                                      params['learning_rate'], params['main_task_weight'])
    base = join(model_save_path, 'ctc', params['label_type_main'] + '_' + params['label_type_sub'],
                params['train_data_size'], model.name)
    model.save_path = _bcast(new_run_directory(base, config_path) if rank == 0 else None)
    stdout = sys.stdout
    if log_to_file and rank == 0:
        sys.stdout = open(join(model.save_path, 'train.log'), 'w')
    try:
        result = do_train(model, params, rank, world)
    finally:
        if log_to_file and rank == 0:
            sys.stdout.close()
            sys.stdout = stdout
    result.update(save_path=model.save_path, rank=rank, world=world, model=model)
    return result


if __name__ == '__main__':
    args = sys.argv
    if len(args) != 3:
        raise ValueError('Length of args should be 3.')
    main(config_path=args[1], model_save_path=args[2])
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
