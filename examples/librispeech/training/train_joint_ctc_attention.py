#!/usr/bin/env python
"""Train the joint CTC-attention model, data parallel (BASELINE configs[3]: 5x512 BLSTM encoder + location attention,
lambda-weighted CTC head, one process per GPU).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/librispeech/training/train_joint_ctc_attention.py <config.yml> <model_save_path>

The reference's multi-GPU attention recipe (examples/csj/training/train_attention.py:60-200) builds N towers in one
graph; here every rank draws the same global batch (batch_size x N utterances, one global max length, np.array_split),
runs the joint model on shard [rank], clips per variable on the tower, averages gradients over towers (RCCL) and
applies the identical update (multi_gpu.tower_step_with).  Rank 0 scores the dev sets with the inference decoder,
checkpoints on a new best and keeps the run directory."""
import os
import random
import sys
import time
from os.path import abspath, dirname, isfile, join

import numpy as np
import torch
import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.librispeech.data.load_dataset_joint_ctc_attention import Dataset                               # noqa: E402
from examples.librispeech.training.train_ctc import _bcast                                                   # noqa: E402
from examples.timit.metrics.attention import do_eval_cer                                                     # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                         # noqa: E402
from examples.timit.training._common import new_run_directory                                                # noqa: E402
from examples.timit.training.train_attention import attention_ler, model_kwargs, run_name                    # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.attention.joint_ctc_attention import JointCTCAttention  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor         # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu                               # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, sync_point                        # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller     # noqa: E402


def do_train(model, params, rank, world):
    map_dir = params.get('map_dir') or join(model.save_path, 'mapping_files')
    if rank == 0 and not isfile(join(map_dir, 'character.txt')):
        write_mapping_files(map_dir)
    if multi_gpu.is_distributed():
        torch.distributed.barrier()                           # the map files exist before any rank reads them
    map_file = join(map_dir, params['label_type'] + '.txt')
    kw = dict(train_data_size=params['train_data_size'], label_type=params['label_type'],
              batch_size=params['batch_size'], map_file_path=map_file, splice=params['splice'],
              num_stack=params['num_stack'], num_skip=params['num_skip'], dataset_root=params.get('dataset_root'))
    seed = params.get('seed', 0)
    train_data = Dataset(data_type='train', max_epoch=params['num_epoch'], sort_utt=True,
                         sort_stop_epoch=params['sort_stop_epoch'], num_gpu=world, **kw)
    train_data.rng = random.Random(seed)                     # identical global batches on every rank
    if rank == 0:
        dev_sets = {}
        for i, name in enumerate(('dev_clean', 'dev_other')):
            dev_sets[name] = Dataset(data_type=name, shuffle=False, **kw)
            dev_sets[name].rng = random.Random(seed + 1 + i)
    small = params['train_data_size'] in ['train100h', 'train460h']

    multi_gpu.broadcast_parameters(model.store)
    optimizer = model._set_optimizer(params['optimizer'], params['learning_rate'])
    model.optimizer = optimizer
    lr_controller = Controller(learning_rate_init=params['learning_rate'],
                               decay_start_epoch=params['decay_start_epoch'], decay_rate=params['decay_rate'],
                               decay_patient_epoch=params['decay_patient_epoch'], lower_better=True)
    saver = Saver(max_to_keep=None)
    kp = [1 - float(params[k]) for k in ('dropout_encoder', 'dropout_decoder', 'dropout_embedding')]
    learning_rate = float(params['learning_rate'])
    print_step = max(1, int(params['print_step'] / world))
    best, not_improved = 1, 0
    result = dict(metric_dev=[], checkpoints=[], steps=0, losses=[])
    start_step = time.time()
    for step, (data, is_new_epoch) in enumerate(train_data):
        inputs, att_labels, ctc_labels, inputs_seq_len, att_labels_seq_len, _ = data
        x, ya, yc, sl, la = inputs[rank], att_labels[rank], ctc_labels[rank], inputs_seq_len[rank], \
            att_labels_seq_len[rank]

        def loss_fn():
            if len(x) == 0:
                return None
            return model.compute_loss(x, ya, list2sparsetensor(yc, padded_value=-1), sl, la, *kp)[0]
        loss = multi_gpu.tower_step_with(model, optimizer, loss_fn, learning_rate=learning_rate)
        result['steps'] = step + 1
        result['losses'].append(float(loss))

        if (step + 1) % print_step == 0:
            ler = torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
            if len(x):
                _, _, _, out_train, out_infer = model.compute_loss(x, ya, list2sparsetensor(yc, padded_value=-1), sl,
                                                                   la, 1.0, 1.0, 1.0, is_training=False)
                ids = model.decode(out_train, out_infer)[1]
                ids = np.asarray(ids.cpu() if hasattr(ids, 'cpu') else ids)
                ler = ler + float(attention_ler(model, ya, la, ids))
            ler = float(multi_gpu.average_scalar(ler))
            if rank == 0:
                print('Step %d (epoch: %.3f): loss = %.3f / ler = %.3f / lr = %.5f (%.3f min)' %
                      (step + 1, train_data.epoch_detail, float(loss), ler, learning_rate,
                       (time.time() - start_step) / 60))
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
                    scores = {}
                    for name, data_set in dev_sets.items():
                        cer, wer = do_eval_cer(session=None, decode_op=None, model=model, dataset=data_set,
                                               label_type=params['label_type'], eval_batch_size=1, map_dir=map_dir,
                                               is_jointctcatt=True)
                        scores[name] = cer
                        print('  CER (%s): %f %%' % (name, cer * 100))
                        print('  WER (%s): %f %%' % (name, wer * 100))
                    metric = scores['dev_clean'] if small else scores['dev_other']
                    result['metric_dev'].append(metric)
                    if metric < best:
                        best, not_improved = metric, 0
                        print('■■■ ↑Best Score (CER)↑ ■■■')
                        path = saver.save(model, join(model.save_path, 'model.ckpt'), global_step=train_data.epoch)
                        result['checkpoints'].append(path)
                        print('Model saved in file: %s' % path)
                    else:
                        not_improved += 1
                    stop = not_improved == params['not_improved_patient_epoch']
                    if not stop:
                        learning_rate = lr_controller.decay_lr(learning_rate=learning_rate, epoch=train_data.epoch,
                                                               value=metric)
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
    params['num_classes'] = {'character': 28, 'character_capital_divide': 72}[params['label_type']]
    params['device'] = device
    model = JointCTCAttention(lambda_weight=params['lambda_weight'], seed=params.get('seed', 0),
                              **model_kwargs(params))
    model.name = run_name(params) + '_lambda' + str(params['lambda_weight'])
    base = join(model_save_path, 'joint_ctc_attention', params['label_type'], params['train_data_size'], model.name)
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
