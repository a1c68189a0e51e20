#!/usr/bin/env python
"""Train the multitask CTC model on TIMIT -- the recipe of examples/timit/training/train_multitask_ctc.py:
main task = characters (CER, the early-stopping metric), sub task = phones (PER on 39 phones).

    python examples/timit/training/train_multitask_ctc.py <config.yml> <model_save_path>"""
import sys
from os.path import abspath, dirname, isfile, join

import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.timit.data.load_dataset_multitask_ctc import Dataset                                           # noqa: E402
from examples.timit.metrics.ctc import do_eval_per, do_eval_cer                                              # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                         # noqa: E402
from examples.timit.training._common import NUM_CLASSES, new_run_directory, run_with_log, training_loop       # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.ctc.multitask_ctc import MultitaskCTC                  # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import list2sparsetensor         # noqa: E402


def do_train(model, params):
    map_dir = params.get('map_dir') or join(model.save_path, 'mapping_files')
    if not isfile(join(map_dir, 'phone2phone.txt')):
        write_mapping_files(map_dir)
    kw = dict(splice=params['splice'], num_stack=params['num_stack'], num_skip=params['num_skip'],
              dataset_root=params.get('dataset_root'))
    lt_main, lt_sub = params['label_type_main'], params['label_type_sub']
    train_data = Dataset(data_type='train', label_type_main=lt_main, label_type_sub=lt_sub,
                         batch_size=params['batch_size'], max_epoch=params['num_epoch'], sort_utt=True,
                         sort_stop_epoch=params['sort_stop_epoch'], **kw)
    dev_data = Dataset(data_type='dev', label_type_main=lt_main, label_type_sub=lt_sub,
                       batch_size=params['batch_size'], sort_utt=False, **kw)
    test_data = Dataset(data_type='test', label_type_main=lt_main, label_type_sub='phone39', batch_size=1,
                        sort_utt=False, **kw)
    keep_prob = 1 - float(params['dropout'])

    def sparse(data, padded):
        inputs, labels_main, labels_sub, inputs_seq_len, _ = data
        return (inputs[0], list2sparsetensor(labels_main[0], padded_value=padded),
                list2sparsetensor(labels_sub[0], padded_value=padded), inputs_seq_len[0])

    def train_step(data, learning_rate):
        x, ym, ys, sl = sparse(data, train_data.padded_value)
        loss, _, _ = model.compute_loss(x, ym, ys, sl, keep_prob)
        model.train(loss, optimizer=params['optimizer'], learning_rate=learning_rate)

    def monitor(data):
        x, ym, ys, sl = sparse(data, -1)
        loss, lm, ls = model.compute_loss(x, ym, ys, sl, 1.0, is_training=False)
        dm, ds = model.decoder(lm, ls, sl, beam_width=params['beam_width'])
        ler_main, ler_sub = model.compute_ler(dm, ds, ym, ys)
        print('  sub-task ler = %.3f' % ler_sub)
        return float(loss), ler_main

    def evaluate(is_test):
        ds_ = test_data if is_test else dev_data
        ev = dict(session=None, decode_op=None, model=model, dataset=ds_, is_test=is_test, eval_batch_size=1,
                  map_dir=map_dir, beam_width=params['beam_width'], is_multitask=True)
        cer, wer = do_eval_cer(label_type=lt_main, **ev)
        print('  WER (main): %f %%' % (wer * 100))
        # the sub task's test labels are stored as index arrays on every set (only the main transcript is a string)
        per = do_eval_per(per_op=None, label_type=lt_sub, **dict(ev, is_test=False))
        print('  PER (sub): %f %%' % (per * 100))
        return cer

    return training_loop(model, params, train_data, dev_data, train_step, monitor, evaluate, 'CER')


def main(config_path, model_save_path, log_to_file=True):
    with open(config_path, 'r') as f:
        params = yaml.safe_load(f)['param']
    params['num_classes_main'] = NUM_CLASSES[params['label_type_main']]
    params['num_classes_sub'] = NUM_CLASSES[params['label_type_sub']]
    model = MultitaskCTC(encoder_type=params['encoder_type'], input_size=params['input_size'],
                         num_units=params['num_units'], num_layers_main=params['num_layers_main'],
                         num_layers_sub=params['num_layers_sub'], num_classes_main=params['num_classes_main'],
                         num_classes_sub=params['num_classes_sub'], main_task_weight=params['main_task_weight'],
                         lstm_impl=params['lstm_impl'], use_peephole=params['use_peephole'], splice=params['splice'],
                         parameter_init=params['weight_init'], clip_grad_norm=params['clip_grad_norm'],
                         clip_activation=params['clip_activation'], num_proj=params['num_proj'],
                         weight_decay=params['weight_decay'], dtype=params.get('dtype', 'bf16'),
                         device=params.get('device', 'cuda:0'))
    model.name += '_' + str(params['num_units']) + '_main' + str(params['num_layers_main'])
    model.name += '_sub' + str(params['num_layers_sub']) + '_' + params['optimizer']
    model.name += '_lr' + str(params['learning_rate']) + '_w' + str(params['main_task_weight'])
    if params['dropout'] != 0:
        model.name += '_drop' + str(params['dropout'])
    model.save_path = new_run_directory(
        join(model_save_path, 'ctc', params['label_type_main'] + '_' + params['label_type_sub'], model.name),
        config_path)
    result = run_with_log(lambda: do_train(model, params), model.save_path, log_to_file)
    result.update(save_path=model.save_path, model=model)
    return result


if __name__ == '__main__':
    args = sys.argv
    if len(args) != 3:
        raise ValueError('Length of args should be 3.')
    main(config_path=args[1], model_save_path=args[2])
