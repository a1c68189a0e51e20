"""PER / CER+WER evaluation loops for the attention models -- mirror of examples/timit/metrics/attention.py:19-245
(do_eval_per, do_eval_cer).  Decoding is the model's greedy inference decoder (dynamic_decode with
GreedyEmbeddingHelper); the hypothesis is cut at the first <EOS> ('>'), the reference at its <SOS>/<EOS> frame
(labels[1 : len-1]).  `session` / `decode_op` / `per_op` are accepted for call compatibility and ignored."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tensorflow_end2end_speech_recognition_amd.utils.io.labels.phone import Idx2phone                 # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.io.labels.character import Idx2char              # noqa: E402
from tensorflow_end2end_speech_recognition_amd.utils.evaluation.edit_distance import (                # noqa: E402
    compute_per, compute_cer, compute_wer)
from examples.timit.metrics.mapping import Map2phone39                                                # noqa: E402


def _unpack(data, is_multitask, is_jointctcatt):
    if is_multitask:
        inputs, _, labels_true, inputs_seq_len, labels_seq_len, _ = data
    elif is_jointctcatt:
        inputs, labels_true, _, inputs_seq_len, labels_seq_len, _ = data
    else:
        inputs, labels_true, inputs_seq_len, labels_seq_len, _ = data
    return inputs, labels_true, inputs_seq_len, labels_seq_len


def _infer(model, inputs, inputs_seq_len):
    """Greedy inference ids [B, <= max_decode_length] (the decode_op_infer of the reference's drivers)."""
    return np.asarray(model.infer(inputs, inputs_seq_len))


def do_eval_per(session, decode_op, per_op, model, dataset, label_type, is_test=False, eval_batch_size=None,
                progressbar=False, is_multitask=False, is_jointctcatt=False, map_dir=None):
    map_dir = map_dir or '../metrics/mapping_files'
    batch_size_original = dataset.batch_size
    dataset.reset()
    if eval_batch_size is not None:
        dataset.batch_size = eval_batch_size
    eval_label_type = dataset.label_type_sub if is_multitask else dataset.label_type
    idx2phone_train = Idx2phone(os.path.join(map_dir, label_type + '.txt'))
    idx2phone_eval = Idx2phone(os.path.join(map_dir, eval_label_type + '.txt'))
    p2p = os.path.join(map_dir, 'phone2phone.txt')
    p2p = p2p if os.path.isfile(p2p) else None
    to39_train, to39_eval = Map2phone39(label_type, p2p), Map2phone39(eval_label_type, p2p)
    per_sum = 0.0
    for data, is_new_epoch in dataset:
        inputs, labels_true, inputs_seq_len, labels_seq_len = _unpack(data, is_multitask, is_jointctcatt)
        labels_pred = _infer(model, inputs[0], inputs_seq_len[0])
        for b in range(inputs[0].shape[0]):
            str_pred = idx2phone_train(np.asarray(labels_pred[b])).split('>')[0].rstrip(' ')
            pred = [p for p in str_pred.split(' ') if p not in ('', '<')]
            if is_test:
                true = labels_true[0][b][0].split(' ')
            else:
                true = idx2phone_eval(np.asarray(labels_true[0][b][1:labels_seq_len[0][b] - 1])).split(' ')
            pred, true = to39_train(pred), to39_eval(true)
            per_sum += compute_per(ref=true, hyp=pred, normalize=True)                 # (:121-123)
        if is_new_epoch:
            break
    if eval_batch_size is not None:
        dataset.batch_size = batch_size_original
    return per_sum / len(dataset)


def do_eval_cer(session, decode_op, model, dataset, label_type, is_test=False, eval_batch_size=None,
                progressbar=False, is_multitask=False, is_jointctcatt=False, map_dir=None):
    """-> (mean CER, mean WER)."""
    map_dir = map_dir or '../metrics/mapping_files'
    batch_size_original = dataset.batch_size
    dataset.reset()
    if eval_batch_size is not None:
        dataset.batch_size = eval_batch_size
    if label_type == 'character':
        idx2char = Idx2char(os.path.join(map_dir, 'character.txt'))
    elif label_type == 'character_capital_divide':
        idx2char = Idx2char(os.path.join(map_dir, 'character_capital_divide.txt'), capital_divide=True, space_mark='_')
    else:
        raise ValueError('label_type must be character or character_capital_divide')
    cer_sum = wer_sum = 0.0
    for data, is_new_epoch in dataset:
        inputs, labels_true, inputs_seq_len, labels_seq_len = _unpack(data, is_multitask, is_jointctcatt)
        labels_pred = _infer(model, inputs[0], inputs_seq_len[0])
        for b in range(inputs[0].shape[0]):
            if is_test:
                str_true = labels_true[0][b][0]
            else:
                str_true = idx2char(np.asarray(labels_true[0][b][1:labels_seq_len[0][b] - 1]))
            str_pred = idx2char(np.asarray(labels_pred[b])).split('>')[0].replace('<', '')
            str_pred = re.sub(r'[_]+', '_', str_pred)
            str_true = re.sub(r'[\'\":;!?,.-]+', '', str_true)
            str_pred = re.sub(r'[\'\":;!?,.-]+', '', str_pred)
            wer_sum += compute_wer(hyp=str_pred.split('_'), ref=str_true.split('_'), normalize=True)
            cer_sum += compute_cer(str_pred=re.sub(r'[_]+', '', str_pred), str_true=re.sub(r'[_]+', '', str_true),
                                   normalize=True)
        if is_new_epoch:
            break
    if eval_batch_size is not None:
        dataset.batch_size = batch_size_original
    return cer_sum / len(dataset), wer_sum / len(dataset)
