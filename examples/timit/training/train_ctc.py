#!/usr/bin/env python
"""Train the CTC model on TIMIT -- the recipe of examples/timit/training/train_ctc.py:38-389 on the MI355X path.

    python examples/timit/training/train_ctc.py <config.yml> <model_save_path>

Same flow: sorted-then-shuffled training batches -> compute_loss / train; every `print_step` steps loss and label
error rate on the current training batch and on one dev batch; at each epoch end (from `eval_start_epoch`) PER (39
phones) or CER/WER on dev, a checkpoint `model.ckpt-<epoch>` and a test-set evaluation on a new best, early stop
after `not_improved_patient_epoch` epochs without one, learning-rate decay by the Controller; `config.yml`,
`train.log`, `complete.txt` in the run directory named like the reference's (:338-371).

What changed with the backend: no graph / session / placeholders (the ops of :69-90 are direct calls), TensorBoard
summaries and matplotlib plots (:105-106,163-167) become `loss.csv` / `ler.csv`.  Extra config keys, all optional:
`dtype` (bf16 | f32), `dataset_root` (else $TIMIT_DATASET_ROOT, else the reference's site paths), `map_dir`
(mapping files are generated there if absent), `device_assembly` (stack / splice on the GPU)."""
import os
import shutil
import sys
import time
from os.path import abspath, dirname, isfile, join

import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.timit.data.load_dataset_ctc import Dataset                                                    # noqa: E402
from examples.timit.metrics.ctc import do_eval_per, do_eval_cer                                            # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                       # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.ctc import CTC                                    # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor        # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.parameter import count_total_parameters                # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, sync_point                       # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller    # noqa: E402

NUM_CLASSES = {'phone61': 61, 'phone48': 48, 'phone39': 39, 'character': 28, 'character_capital_divide': 72}


def _prepare(model, data, params):
    """One batch of the iterator -> what compute_loss takes (optionally assembled on the device)."""
    inputs, labels, inputs_seq_len, _ = data
    x, sl = inputs[0], inputs_seq_len[0]
    if params.get('device_assembly'):
        from tensorflow_end2end_speech_recognition_amd.utils.io.inputs.device import assemble
        x, sl = assemble(x, sl, params['num_stack'], params['num_skip'], params['splice'], device=model.device)
    return x, labels[0], sl


def do_train(model, params):
    root, map_dir = params.get('dataset_root'), params.get('map_dir') or join(model.save_path, 'mapping_files')
    if not isfile(join(map_dir, 'phone2phone.txt')):
        write_mapping_files(map_dir)
    kw = dict(splice=params['splice'], num_stack=params['num_stack'], num_skip=params['num_skip'], dataset_root=root,
              device_assembly=bool(params.get('device_assembly')))
    train_data = Dataset(data_type='train', label_type=params['label_type'], batch_size=params['batch_size'],
                         max_epoch=params['num_epoch'], sort_utt=True, sort_stop_epoch=params['sort_stop_epoch'], **kw)
    dev_data = Dataset(data_type='dev', label_type=params['label_type'], batch_size=params['batch_size'],
                       sort_utt=False, **kw)
    is_char = 'char' in params['label_type']
    # the test set is scored on 39 phones whatever the training set was (:56-68)
    test_data = Dataset(data_type='test', label_type=params['label_type'] if is_char else 'phone39', batch_size=1,
                        sort_utt=False, **kw)

    lr_controller = Controller(learning_rate_init=params['learning_rate'],
                               decay_start_epoch=params['decay_start_epoch'], decay_rate=params['decay_rate'],
                               decay_patient_epoch=params['decay_patient_epoch'], lower_better=True)
    saver = Saver(max_to_keep=None)
    parameters_dict, total_parameters = count_total_parameters(model.store.state_dict())
    for name in sorted(parameters_dict.keys()):
        print('%s %d' % (name, parameters_dict[name]))
    print('Total %d variables, %s M parameters' % (len(parameters_dict), '{:,}'.format(total_parameters / 1000000)))

    csv_loss = open(join(model.save_path, 'loss.csv'), 'w')
    csv_ler = open(join(model.save_path, 'ler.csv'), 'w')
    csv_loss.write('step,train,dev\n')
    csv_ler.write('step,train,dev\n')
    start_time_train = start_time_epoch = start_time_step = time.time()
    ler_dev_best, not_improved_epoch = 1, 0
    learning_rate = float(params['learning_rate'])
    keep_prob = 1 - float(params['dropout'])
    result = dict(ler_dev=[], ler_test=None, checkpoints=[])
    for step, (data, is_new_epoch) in enumerate(train_data):
        x, labels, sl = _prepare(model, data, params)
        labels_st = list2sparsetensor(labels, padded_value=train_data.padded_value)
        loss, _ = model.compute_loss(x, labels_st, sl, keep_prob)
        model.train(loss, optimizer=params['optimizer'], learning_rate=learning_rate)

        if (step + 1) % params['print_step'] == 0:
            xd, labels_d, sld = _prepare(model, next(dev_data)[0], params)
            labels_d_st = list2sparsetensor(labels_d, padded_value=dev_data.padded_value)
            loss_train, logits_train = model.compute_loss(x, labels_st, sl, 1.0, is_training=False)
            loss_dev, logits_dev = model.compute_loss(xd, labels_d_st, sld, 1.0, is_training=False)
            ler_train = model.compute_ler(model.decoder(logits_train, sl, params['beam_width']), labels_st)
            ler_dev = model.compute_ler(model.decoder(logits_dev, sld, params['beam_width']), labels_d_st)
            csv_loss.write('%d,%f,%f\n' % (step, float(loss_train), float(loss_dev)))
            csv_ler.write('%d,%f,%f\n' % (step, ler_train, ler_dev))
            print('Step %d (epoch: %.3f): loss = %.3f (%.3f) / ler = %.3f (%.3f) / lr = %.5f (%.3f min)' %
                  (step + 1, train_data.epoch_detail, float(loss_train), float(loss_dev), ler_train, ler_dev,
                   learning_rate, (time.time() - start_time_step) / 60))
            sys.stdout.flush()
            start_time_step = time.time()

        if is_new_epoch:
            sync_point()          # pending asynchronous error checks of this epoch's steps are raised here
            print('-----EPOCH:%d (%.3f min)-----' % (train_data.epoch, (time.time() - start_time_epoch) / 60))
            csv_loss.flush()
            csv_ler.flush()
            if train_data.epoch >= params['eval_start_epoch']:
                start_time_eval = time.time()
                print('=== Dev Data Evaluation ===')
                ev = dict(model=model, label_type=params['label_type'], eval_batch_size=1, map_dir=map_dir,
                          beam_width=params['beam_width'])
                if is_char:
                    ler_dev_epoch, wer_dev_epoch = do_eval_cer(session=None, decode_op=None, dataset=dev_data, **ev)
                    print('  CER: %f %%' % (ler_dev_epoch * 100))
                    print('  WER: %f %%' % (wer_dev_epoch * 100))
                else:
                    ler_dev_epoch = do_eval_per(session=None, decode_op=None, per_op=None, dataset=dev_data, **ev)
                    print('  PER: %f %%' % (ler_dev_epoch * 100))
                result['ler_dev'].append(ler_dev_epoch)
                if ler_dev_epoch < ler_dev_best:
                    ler_dev_best, not_improved_epoch = ler_dev_epoch, 0
                    print('■■■ ↑Best Score (%s)↑ ■■■' % ('CER' if is_char else 'PER'))
                    save_path = saver.save(model, join(model.save_path, 'model.ckpt'), global_step=train_data.epoch)
                    result['checkpoints'].append(save_path)
                    print('Model saved in file: %s' % save_path)
                    print('=== Test Data Evaluation ===')
                    if is_char:
                        ler_test, wer_test = do_eval_cer(session=None, decode_op=None, dataset=test_data,
                                                         is_test=True, **ev)
                        print('  CER: %f %%' % (ler_test * 100))
                        print('  WER: %f %%' % (wer_test * 100))
                    else:
                        ler_test = do_eval_per(session=None, decode_op=None, per_op=None, dataset=test_data,
                                               is_test=True, **ev)
                        print('  PER: %f %%' % (ler_test * 100))
                    result['ler_test'] = ler_test
                else:
                    not_improved_epoch += 1
                print('Evaluation time: %.3f min' % ((time.time() - start_time_eval) / 60))
                if not_improved_epoch == params['not_improved_patient_epoch']:
                    break
                learning_rate = lr_controller.decay_lr(learning_rate=learning_rate, epoch=train_data.epoch,
                                                       value=ler_dev_epoch)
            start_time_epoch = time.time()

    print('Total time: %.3f hour' % ((time.time() - start_time_train) / 3600))
    csv_loss.close()
    csv_ler.close()
    sync_point()
    with open(join(model.save_path, 'complete.txt'), 'w') as f:       # marks the run directory as used (:304-305)
        f.write('')
    return result


def build_model(params):
    if params['label_type'] not in NUM_CLASSES:
        raise TypeError
    params['num_classes'] = NUM_CLASSES[params['label_type']]
    model = CTC(encoder_type=params['encoder_type'], input_size=params['input_size'], splice=params['splice'],
                num_stack=params['num_stack'], num_units=params['num_units'], num_layers=params['num_layers'],
                num_classes=params['num_classes'], lstm_impl=params['lstm_impl'],
                use_peephole=params['use_peephole'], parameter_init=params['weight_init'],
                clip_grad_norm=params['clip_grad_norm'], clip_activation=params['clip_activation'],
                num_proj=params['num_proj'], weight_decay=params['weight_decay'],
                dtype=params.get('dtype', 'bf16'), device=params.get('device', 'cuda:0'))
    # run-directory name, as :338-351
    model.name += '_' + str(params['num_units'])
    model.name += '_' + str(params['num_layers'])
    model.name += '_' + params['optimizer']
    model.name += '_lr' + str(params['learning_rate'])
    if params['num_proj'] not in (0, None):
        model.name += '_proj' + str(params['num_proj'])
    if params['dropout'] != 0:
        model.name += '_drop' + str(params['dropout'])
    if params['num_stack'] != 1:
        model.name += '_stack' + str(params['num_stack'])
    if params['weight_decay'] != 0:
        model.name += '_wd' + str(params['weight_decay'])
    return model


def main(config_path, model_save_path, log_to_file=True):
    with open(config_path, 'r') as f:
        params = yaml.safe_load(f)['param']
    model = build_model(params)
    base = join(model_save_path, 'ctc', params['label_type'], model.name)
    # never reuse a directory that holds a finished or a started run (:356-368)
    new_model_path, model_index = base, 0
    while isfile(join(new_model_path, 'complete.txt')) or isfile(join(new_model_path, 'config.yml')):
        model_index += 1
        new_model_path = base + '_' + str(model_index)
    os.makedirs(new_model_path, exist_ok=True)
    model.save_path = new_model_path
    shutil.copyfile(config_path, join(model.save_path, 'config.yml'))
    stdout = sys.stdout
    if log_to_file:
        sys.stdout = open(join(model.save_path, 'train.log'), 'w')
    try:
        result = do_train(model=model, params=params)
    finally:
        if log_to_file:
            sys.stdout.close()
            sys.stdout = stdout
    result.update(save_path=model.save_path, model=model)
    return result


if __name__ == '__main__':
    args = sys.argv
    if len(args) != 3:
        raise ValueError('Length of args should be 3.')
    main(config_path=args[1], model_save_path=args[2])
