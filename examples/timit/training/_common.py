"""What the four TIMIT training drivers of the reference repeat verbatim (examples/timit/training/train_*.py):
run-directory bookkeeping (:338-371 of train_ctc.py) and the step / print_step / epoch-evaluation / checkpoint /
early-stop / learning-rate loop (:108-305), parameterised by three callables the model family supplies."""
import os
import shutil
import sys
import time
from os.path import isfile, join

from tensorflow_end2end_speech_recognition_amd.utils.parameter import count_total_parameters
from tensorflow_end2end_speech_recognition_amd.utils.training.checkpoint import Saver, sync_point
from tensorflow_end2end_speech_recognition_amd.utils.training.learning_rate_controller import Controller

NUM_CLASSES = {'phone61': 61, 'phone48': 48, 'phone39': 39, 'character': 28, 'character_capital_divide': 72}


def new_run_directory(base, config_path):
    """Never reuse a directory that holds a finished (complete.txt) or started (config.yml) run."""
    path, index = base, 0
    while isfile(join(path, 'complete.txt')) or isfile(join(path, 'config.yml')):
        index += 1
        path = base + '_' + str(index)
    os.makedirs(path, exist_ok=True)
    shutil.copyfile(config_path, join(path, 'config.yml'))
    return path


def run_with_log(fn, save_path, log_to_file=True):
    stdout = sys.stdout
    if log_to_file:
        sys.stdout = open(join(save_path, 'train.log'), 'w')
    try:
        return fn()
    finally:
        if log_to_file:
            sys.stdout.close()
            sys.stdout = stdout


def training_loop(model, params, train_data, dev_data, train_step, monitor, evaluate, metric_name):
    """train_step(data, learning_rate) -> None; monitor(data) -> (loss, ler) without dropout;
    evaluate(is_test) -> the epoch metric (PER or CER; lower is better).  Returns the run summary."""
    lr_controller = Controller(learning_rate_init=params['learning_rate'],
                               decay_start_epoch=params['decay_start_epoch'], decay_rate=params['decay_rate'],
                               decay_patient_epoch=params['decay_patient_epoch'], lower_better=True)
    saver = Saver(max_to_keep=None)
    parameters_dict, total_parameters = count_total_parameters(model.store.state_dict())
    for name in sorted(parameters_dict.keys()):
        print('%s %d' % (name, parameters_dict[name]))
    print('Total %d variables, %s M parameters' % (len(parameters_dict), '{:,}'.format(total_parameters / 1000000)))
    csv = open(join(model.save_path, 'loss_ler.csv'), 'w')
    csv.write('step,loss_train,loss_dev,ler_train,ler_dev\n')
    start_time_train = start_time_epoch = start_time_step = time.time()
    ler_dev_best, not_improved_epoch = 1, 0
    learning_rate = float(params['learning_rate'])
    result = dict(ler_dev=[], ler_test=None, checkpoints=[], steps=0)
    for step, (data, is_new_epoch) in enumerate(train_data):
        train_step(data, learning_rate)
        result['steps'] = step + 1
        if (step + 1) % params['print_step'] == 0:
            loss_train, ler_train = monitor(data)
            loss_dev, ler_dev = monitor(next(dev_data)[0])
            csv.write('%d,%f,%f,%f,%f\n' % (step, loss_train, loss_dev, ler_train, ler_dev))
            print('Step %d (epoch: %.3f): loss = %.3f (%.3f) / ler = %.3f (%.3f) / lr = %.5f (%.3f min)' %
                  (step + 1, train_data.epoch_detail, loss_train, loss_dev, ler_train, ler_dev, learning_rate,
                   (time.time() - start_time_step) / 60))
            sys.stdout.flush()
            start_time_step = time.time()
        if is_new_epoch:
            sync_point()          # pending asynchronous error checks of this epoch's steps are raised here
            print('-----EPOCH:%d (%.3f min)-----' % (train_data.epoch, (time.time() - start_time_epoch) / 60))
            csv.flush()
            if train_data.epoch >= params['eval_start_epoch']:
                start_time_eval = time.time()
                print('=== Dev Data Evaluation ===')
                ler_dev_epoch = evaluate(False)
                print('  %s: %f %%' % (metric_name, ler_dev_epoch * 100))
                result['ler_dev'].append(ler_dev_epoch)
                if ler_dev_epoch < ler_dev_best:
                    ler_dev_best, not_improved_epoch = ler_dev_epoch, 0
                    print('■■■ ↑Best Score (%s)↑ ■■■' % metric_name)
                    save_path = saver.save(model, join(model.save_path, 'model.ckpt'), global_step=train_data.epoch)
                    result['checkpoints'].append(save_path)
                    print('Model saved in file: %s' % save_path)
                    print('=== Test Data Evaluation ===')
                    result['ler_test'] = evaluate(True)
                    print('  %s: %f %%' % (metric_name, result['ler_test'] * 100))
                else:
                    not_improved_epoch += 1
                print('Evaluation time: %.3f min' % ((time.time() - start_time_eval) / 60))
                if not_improved_epoch == params.get('not_improved_patient_epoch', -1):
                    break
                learning_rate = lr_controller.decay_lr(learning_rate=learning_rate, epoch=train_data.epoch,
                                                       value=ler_dev_epoch)
            start_time_epoch = time.time()
    print('Total time: %.3f hour' % ((time.time() - start_time_train) / 3600))
    csv.close()
    sync_point()
    with open(join(model.save_path, 'complete.txt'), 'w') as f:
        f.write('')
    return result
