#!/usr/bin/env python
"""Train the CTC model on Librispeech, data parallel -- the recipe of
examples/librispeech/training/train_ctc.py:30-470 on the MI355X path.

    python examples/librispeech/training/train_ctc.py <config.yml> <model_save_path>                    # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/librispeech/training/train_ctc.py <config.yml> <model_save_path>                        # N GPUs

The reference runs N in-graph towers in one process (`gpu_indices` argument); here there is one process per GPU
(RCCL over xGMI) and the same arithmetic: every rank draws the SAME global batch of batch_size * N utterances
(identically seeded private samplers), padded to the global max length and split with np.array_split, and works on shard
[rank]; gradients are clipped per variable on the tower, averaged over towers, and every rank applies the identical
update (utils/training/multi_gpu.py tower_step).  Loss / label error rate are tower means (:136-139).  Rank 0 logs,
evaluates (dev_clean + dev_other, then test_clean + test_other on a new best), checkpoints and keeps the run
directory; the other ranks follow its early-stop / learning-rate decisions through a broadcast."""
import os
import random
import shutil
import sys
import time
from os.path import abspath, dirname, isfile, join

import numpy as np
import torch
import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.librispeech.data.load_dataset_ctc import Dataset                                              # noqa: E402
from examples.librispeech.metrics.ctc import do_eval_cer, do_eval_wer                                       # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                        # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC                                    # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor        # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.parameter import count_total_parameters                # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training import multi_gpu                              # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, sync_point                       # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller    # noqa: E402


def _num_classes(params):
    if params['label_type'] == 'character':
        return 28
    if params['label_type'] == 'character_capital_divide':
        return {'train100h': 72, 'train460h': 77, 'train960h': 77}[params['train_data_size']]   # :420-428
    if params['label_type'] == 'word':
        return {'train100h': 7213, 'train460h': 18641, 'train960h': 26642}[params['train_data_size']]
    raise TypeError


def _bcast(value, src=0):
    """rank 0's decision (float) to every rank."""
    return multi_gpu.broadcast_decision(value, src=src)


def _tower_eval(model, inputs, labels, seq_len, padded_value, beam_width):
    """loss and label error rate of this rank's shard, averaged over towers (:136-139)."""
    if len(inputs) == 0:
        z = torch.zeros((), dtype=torch.float32, device=model.store.flat.device)
        return float(multi_gpu.average_scalar(z)), float(multi_gpu.average_scalar(z))
    st = list2sparsetensor(labels, padded_value=padded_value)
    loss, logits = model.compute_loss(inputs, st, seq_len, 1.0, is_training=False)
    ler = model.compute_ler(model.decoder(logits, seq_len, beam_width), st)
    dev = model.store.flat.device
    return (float(multi_gpu.average_scalar(loss.detach())),
            float(multi_gpu.average_scalar(torch.tensor(float(ler), dtype=torch.float32, device=dev))))


def do_train(model, params, rank, world):
    root = params.get('dataset_root')
    map_dir = params.get('map_dir') or join(model.save_path, 'mapping_files')
    if rank == 0 and not isfile(join(map_dir, 'character.txt')):
        write_mapping_files(map_dir)
    kw = dict(train_data_size=params['train_data_size'], label_type=params['label_type'],
              batch_size=params['batch_size'], splice=params['splice'], num_stack=params['num_stack'],
              num_skip=params['num_skip'], dataset_root=root)
    seed = params.get('seed', 0)
    train_data = Dataset(data_type='train', max_epoch=params['num_epoch'], sort_utt=True,
                         sort_stop_epoch=params['sort_stop_epoch'], num_gpu=world, **kw)
    # every rank must draw the SAME global batches: the samplers that all ranks advance together get private,
    # identically seeded generators; the sets rank 0 evaluates alone are separate objects
    train_data.rng = random.Random(seed)
    small = params['train_data_size'] in ['train100h', 'train460h']
    dev_step_data = Dataset(data_type='dev_clean' if small else 'dev_other', shuffle=True, num_gpu=world, **kw)
    dev_step_data.rng = random.Random(seed + 1)
    if rank == 0:
        dev_clean_data = Dataset(data_type='dev_clean', shuffle=True, **kw)
        dev_other_data = Dataset(data_type='dev_other', shuffle=True, **kw)
        test_clean_data = Dataset(data_type='test_clean', shuffle=True, **kw)
        test_other_data = Dataset(data_type='test_other', shuffle=True, **kw)
        for i, d in enumerate((dev_clean_data, dev_other_data, test_clean_data, test_other_data)):
            d.rng = random.Random(seed + 2 + i)

    multi_gpu.broadcast_parameters(model.store)       # identical replicas before the first step
    optimizer = model._set_optimizer(params['optimizer'], params['learning_rate'])
    model.optimizer = optimizer
    lr_controller = Controller(learning_rate_init=params['learning_rate'],
                               decay_start_epoch=params['decay_start_epoch'], decay_rate=params['decay_rate'],
                               decay_patient_epoch=params['decay_patient_epoch'], lower_better=True)
    saver = Saver(max_to_keep=None)
    if rank == 0:
        parameters_dict, total_parameters = count_total_parameters(model.store.state_dict())
        for name in sorted(parameters_dict.keys()):
            print('%s %d' % (name, parameters_dict[name]))
        print('Total %d variables, %s M parameters' % (len(parameters_dict), '{:,}'.format(total_parameters / 1e6)))
        csv = open(join(model.save_path, 'loss_ler.csv'), 'w')
        csv.write('step,loss_train,loss_dev,ler_train,ler_dev\n')
    start_time_train = start_time_epoch = start_time_step = time.time()
    ler_dev_best, not_improved_epoch = 1, 0
    learning_rate = float(params['learning_rate'])
    keep_prob = 1 - float(params['dropout'])
    print_step = max(1, int(params['print_step'] / world))               # :198
    result = dict(metric_dev=[], checkpoints=[], test=None, steps=0)
    for step, (data, is_new_epoch) in enumerate(train_data):
        inputs, labels, inputs_seq_len, _ = data
        x, y, sl = inputs[rank], labels[rank], inputs_seq_len[rank]
        y_st = list2sparsetensor(y, padded_value=train_data.padded_value) if len(x) else None
        multi_gpu.tower_step(model, optimizer, x, y_st, sl, keep_prob, learning_rate=learning_rate)
        result['steps'] = step + 1

        if (step + 1) % print_step == 0:
            dinputs, dlabels, dseq, _ = dev_step_data.next()[0]
            loss_train, ler_train = _tower_eval(model, x, y, sl, train_data.padded_value, params['beam_width'])
            loss_dev, ler_dev = _tower_eval(model, dinputs[rank], dlabels[rank], dseq[rank],
                                            dev_step_data.padded_value, params['beam_width'])
            if rank == 0:
                csv.write('%d,%f,%f,%f,%f\n' % (step, loss_train, loss_dev, ler_train, ler_dev))
                print('Step %d (epoch: %.3f): loss = %.3f (%.3f) / ler = %.3f (%.3f) / lr = %.5f (%.3f min)' %
                      (step + 1, train_data.epoch_detail, loss_train, loss_dev, ler_train, ler_dev, learning_rate,
                       (time.time() - start_time_step) / 60))
                sys.stdout.flush()
            start_time_step = time.time()

        if is_new_epoch:
            sync_point()          # pending asynchronous error checks of this epoch's steps are raised here
            stop = False
            if rank == 0:
                print('-----EPOCH:%d (%.3f min)-----' % (train_data.epoch, (time.time() - start_time_epoch) / 60))
                csv.flush()
            if train_data.epoch >= params['eval_start_epoch']:
                if rank == 0:
                    start_time_eval = time.time()
                    ev = dict(session=None, decode_ops=None, model=model, eval_batch_size=1, map_dir=map_dir,
                              beam_width=params['beam_width'])
                    print('=== Dev Data Evaluation ===')
                    if params['label_type'] == 'word':
                        m_clean = do_eval_wer(dataset=dev_clean_data, train_data_size=params['train_data_size'], **ev)
                        m_other = do_eval_wer(dataset=dev_other_data, train_data_size=params['train_data_size'], **ev)
                        print('  WER (clean): %f %%' % (m_clean * 100))
                        print('  WER (other): %f %%' % (m_other * 100))
                    else:
                        m_clean, w_clean = do_eval_cer(dataset=dev_clean_data, label_type=params['label_type'], **ev)
                        print('  CER (clean): %f %%' % (m_clean * 100))
                        print('  WER (clean): %f %%' % (w_clean * 100))
                        m_other, w_other = do_eval_cer(dataset=dev_other_data, label_type=params['label_type'], **ev)
                        print('  CER (other): %f %%' % (m_other * 100))
                        print('  WER (other): %f %%' % (w_other * 100))
                    metric_epoch = m_clean if small else m_other              # :300-303
                    result['metric_dev'].append(metric_epoch)
                    if metric_epoch < ler_dev_best:
                        ler_dev_best, not_improved_epoch = metric_epoch, 0
                        print('■■■ ↑Best Score↑ ■■■')
                        save_path = saver.save(model, join(model.save_path, 'model.ckpt'),
                                               global_step=train_data.epoch)
                        result['checkpoints'].append(save_path)
                        print('Model saved in file: %s' % save_path)
                        print('=== Test Data Evaluation ===')
                        tv = dict(ev, is_test=True)
                        if params['label_type'] == 'word':
                            t_clean = do_eval_wer(dataset=test_clean_data, train_data_size=params['train_data_size'], **tv)
                            t_other = do_eval_wer(dataset=test_other_data, train_data_size=params['train_data_size'], **tv)
                        else:
                            t_clean, _ = do_eval_cer(dataset=test_clean_data, label_type=params['label_type'], **tv)
                            t_other, _ = do_eval_cer(dataset=test_other_data, label_type=params['label_type'], **tv)
                        print('  error rate (clean): %f %%' % (t_clean * 100))
                        print('  error rate (other): %f %%' % (t_other * 100))
                        result['test'] = (t_clean, t_other)
                    else:
                        not_improved_epoch += 1
                    print('Evaluation time: %.3f min' % ((time.time() - start_time_eval) / 60))
                    stop = not_improved_epoch == params['not_improved_patient_epoch']
                    if not stop:
                        learning_rate = lr_controller.decay_lr(learning_rate=learning_rate, epoch=train_data.epoch,
                                                               value=metric_epoch)
                stop, learning_rate = _bcast((stop, learning_rate))
            if stop:
                break
            start_time_epoch = time.time()

    if rank == 0:
        print('Total time: %.3f hour' % ((time.time() - start_time_train) / 3600))
        csv.close()
        sync_point()
        with open(join(model.save_path, 'complete.txt'), 'w') as f:
            f.write('')
    return result


def build_model(params, device):
    params['num_classes'] = _num_classes(params)
    model = CTC(encoder_type=params['encoder_type'], input_size=params['input_size'], splice=params['splice'],
                num_stack=params['num_stack'], num_units=params['num_units'], num_layers=params['num_layers'],
                num_classes=params['num_classes'], lstm_impl=params['lstm_impl'],
                use_peephole=params['use_peephole'], parameter_init=params['weight_init'],
                clip_grad_norm=params['clip_grad_norm'], clip_activation=params['clip_activation'],
                num_proj=params['num_proj'], weight_decay=params['weight_decay'],
                bottleneck_dim=params.get('bottleneck_dim'), dtype=params.get('dtype', 'bf16'), device=device,
                seed=params.get('seed', 0))
    model.name += '_' + str(params['num_units']) + '_' + str(params['num_layers']) + '_' + params['optimizer']
    model.name += '_lr' + str(params['learning_rate'])
    if params['num_proj'] not in (0, None):
        model.name += '_proj' + str(params['num_proj'])
    if params['dropout'] != 0:
        model.name += '_drop' + str(params['dropout'])
    if params['num_stack'] != 1:
        model.name += '_stack' + str(params['num_stack'])
    if params['weight_decay'] != 0:
        model.name += '_wd' + str(params['weight_decay'])
    if params.get('bottleneck_dim') not in (0, None):
        model.name += '_bottle' + str(params['bottleneck_dim'])
    return model


def main(config_path, model_save_path, log_to_file=True):
    with open(config_path, 'r') as f:
        params = yaml.safe_load(f)['param']
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    device = params.get('device') or 'cuda:%d' % local_rank
    if device.startswith('cuda'):
        torch.cuda.set_device(torch.device(device))
    rank, world = multi_gpu.init_process_group(device)
    model = build_model(params, device)
    base = join(model_save_path, 'ctc', params['label_type'], params['train_data_size'], model.name)
    new_model_path = base
    if rank == 0:
        model_index = 0
        while isfile(join(new_model_path, 'complete.txt')) or isfile(join(new_model_path, 'config.yml')):
            model_index += 1
            new_model_path = base + '_' + str(model_index)
        os.makedirs(new_model_path, exist_ok=True)
        shutil.copyfile(config_path, join(new_model_path, 'config.yml'))
    model.save_path = _bcast(new_model_path)
    stdout = sys.stdout
    if log_to_file and rank == 0:
        sys.stdout = open(join(model.save_path, 'train.log'), 'w')
    try:
        result = do_train(model=model, params=params, rank=rank, world=world)
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
