#!/usr/bin/env python
"""Train the attention model on TIMIT -- the recipe of examples/timit/training/train_attention.py:33-430.

    python examples/timit/training/train_attention.py <config.yml> <model_save_path>

Flow as train_ctc.py (shared loop in _common.py); family-specific: the dataset yields <SOS> y <EOS> targets and
their lengths, compute_loss takes three keep-probabilities, the monitored label error rate and the epoch PER / CER
come from the greedy inference decoder (decode_op_infer, :103-106)."""
import sys
from os.path import abspath, dirname, isfile, join

import numpy as np
import yaml

ROOT = dirname(dirname(dirname(dirname(abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from examples.timit.data.load_dataset_attention import Dataset                                              # noqa: E402
from examples.timit.metrics.attention import do_eval_per, do_eval_cer                                       # noqa: E402
from examples.timit.metrics.mapping_files import write_mapping_files                                        # noqa: E402
from examples.timit.training._common import NUM_CLASSES, new_run_directory, run_with_log, training_loop      # noqa: E402
from tensorflow_end2end_speech_recognition_amd.models.attention.attention_seq2seq import AttentionSeq2Seq   # noqa: E402


def _cut(ids, eos):
    """prediction up to (excluding) the first <EOS>."""
    out = []
    for v in ids:
        if v == eos:
            break
        out.append(int(v))
    return out


def attention_ler(model, labels, labels_seq_len, ids_infer):
    """Label error rate of the greedy inference decode against y of <SOS> y <EOS> (:165-185)."""
    true = [[int(v) for v in labels[b][1:labels_seq_len[b] - 1]] for b in range(len(labels))]
    pred = [_cut(ids_infer[b], model.eos_index) for b in range(len(labels))]
    return model.compute_ler(true, pred)


def model_kwargs(params):
    return dict(
        input_size=params['input_size'] * params['num_stack'], encoder_type=params['encoder_type'],
        encoder_num_units=params['encoder_num_units'], encoder_num_layers=params['encoder_num_layers'],
        encoder_num_proj=params['encoder_num_proj'], attention_type=params['attention_type'],
        attention_dim=params['attention_dim'], decoder_type=params['decoder_type'],
        decoder_num_units=params['decoder_num_units'], decoder_num_layers=params['decoder_num_layers'],
        embedding_dim=params['embedding_dim'], num_classes=params['num_classes'], sos_index=params['num_classes'],
        eos_index=params['num_classes'] + 1, max_decode_length=params['max_decode_length'],
        lstm_impl='LSTMBlockCell', use_peephole=params['use_peephole'], parameter_init=params['weight_init'],
        clip_grad_norm=params['clip_grad_norm'], clip_activation_encoder=params['clip_activation_encoder'],
        clip_activation_decoder=params['clip_activation_decoder'], weight_decay=params['weight_decay'],
        time_major=True, sharpening_factor=params['sharpening_factor'],
        logits_temperature=params['logits_temperature'], sigmoid_smoothing=params['sigmoid_smoothing'],
        dtype=params.get('dtype', 'f32'), device=params.get('device', 'cuda:0'))


def run_name(params):
    """:376-401."""
    name = 'en' + str(params['encoder_num_units']) + '_' + str(params['encoder_num_layers'])
    name += '_att' + str(params['attention_dim'])
    name += '_de' + str(params['decoder_num_units']) + '_' + str(params['decoder_num_layers'])
    name += '_' + params['optimizer'] + '_lr' + str(params['learning_rate']) + '_' + params['attention_type']
    for key, tag in (('dropout_encoder', '_dropen'), ('dropout_decoder', '_dropde'), ('dropout_embedding', '_dropem')):
        if params[key] != 0:
            name += tag + str(params[key])
    if params['num_stack'] != 1:
        name += '_stack' + str(params['num_stack'])
    if params['weight_decay'] != 0:
        name += 'wd' + str(params['weight_decay'])
    if params['sharpening_factor'] != 1:
        name += '_sharp' + str(params['sharpening_factor'])
    if params['logits_temperature'] != 1:
        name += '_temp' + str(params['logits_temperature'])
    return name


def make_datasets(dataset_cls, params, map_dir):
    """:36-70: train / dev on the training label set, test on 39 phones (or characters)."""
    is_char = 'char' in params['label_type']
    map_train = join(map_dir, params['label_type'] + '.txt')
    map_eval = map_train if is_char else join(map_dir, 'phone39.txt')
    kw = dict(splice=params['splice'], num_stack=params['num_stack'], num_skip=params['num_skip'],
              dataset_root=params.get('dataset_root'))
    train = dataset_cls(data_type='train', label_type=params['label_type'], batch_size=params['batch_size'],
                        map_file_path=map_train, max_epoch=params['num_epoch'], sort_utt=True,
                        sort_stop_epoch=params['sort_stop_epoch'], **kw)
    dev = dataset_cls(data_type='dev', label_type=params['label_type'], batch_size=params['batch_size'],
                      map_file_path=map_train, sort_utt=False, **kw)
    test = dataset_cls(data_type='test', label_type=params['label_type'] if is_char else 'phone39', batch_size=1,
                       map_file_path=map_eval, sort_utt=False, **kw)
    return train, dev, test


def do_train(model, params):
    map_dir = params.get('map_dir') or join(model.save_path, 'mapping_files')
    if not isfile(join(map_dir, 'phone2phone.txt')):
        write_mapping_files(map_dir)
    train_data, dev_data, test_data = make_datasets(Dataset, params, map_dir)
    is_char = 'char' in params['label_type']
    kp = [1 - float(params[k]) for k in ('dropout_encoder', 'dropout_decoder', 'dropout_embedding')]

    def train_step(data, learning_rate):
        inputs, labels, inputs_seq_len, labels_seq_len, _ = data
        loss, _, _, _ = model.compute_loss(inputs[0], labels[0], inputs_seq_len[0], labels_seq_len[0], *kp)
        model.train(loss, optimizer=params['optimizer'], learning_rate=learning_rate)

    def monitor(data):
        inputs, labels, inputs_seq_len, labels_seq_len, _ = data
        loss, _, out_train, out_infer = model.compute_loss(inputs[0], labels[0], inputs_seq_len[0],
                                                           labels_seq_len[0], 1.0, 1.0, 1.0, is_training=False)
        _, ids_infer = model.decode(out_train, out_infer)
        ids_infer = np.asarray(ids_infer.cpu() if hasattr(ids_infer, 'cpu') else ids_infer)
        return float(loss), attention_ler(model, labels[0], labels_seq_len[0], ids_infer)

    def evaluate(is_test):
        ev = dict(session=None, decode_op=None, model=model, dataset=test_data if is_test else dev_data,
                  label_type=params['label_type'], is_test=is_test, eval_batch_size=1, map_dir=map_dir)
        if is_char:
            cer, wer = do_eval_cer(**ev)
            print('  WER: %f %%' % (wer * 100))
            return cer
        return do_eval_per(per_op=None, **ev)

    return training_loop(model, params, train_data, dev_data, train_step, monitor, evaluate,
                         'CER' if is_char else 'PER')


def main(config_path, model_save_path, log_to_file=True):
    with open(config_path, 'r') as f:
        params = yaml.safe_load(f)['param']
    if params['label_type'] not in NUM_CLASSES:
        raise TypeError
    params['num_classes'] = NUM_CLASSES[params['label_type']]
    model = AttentionSeq2Seq(**model_kwargs(params))
    model.name = run_name(params)
    model.save_path = new_run_directory(join(model_save_path, 'attention', params['label_type'], model.name),
                                        config_path)
    result = run_with_log(lambda: do_train(model, params), model.save_path, log_to_file)
    result.update(save_path=model.save_path, model=model)
    return result


if __name__ == '__main__':
    args = sys.argv
    if len(args) != 3:
        raise ValueError('Length of args should be 3.')
    main(config_path=args[1], model_save_path=args[2])
