"""Checkpoints with the call shape the recipes use around tf.train.Saver.

Reference call sites: `saver = tf.train.Saver(max_to_keep=None)`
(examples/timit/training/train_ctc.py:100), `saver.save(sess, join(model.save_path, 'model.ckpt'),
global_step=train_data.epoch)` (:220-223), and on the evaluation side
`ckpt = tf.train.get_checkpoint_state(model.save_path)`, `ckpt.model_checkpoint_path`,
`.../model.ckpt-<epoch>`, `saver.restore(sess, model_path)` (examples/timit/evaluation/eval_ctc.py:74-86).

Kept: the `<prefix>-<global_step>` naming, the `checkpoint` index file in the directory (same two
keys, same text format) and the variable names (the TF names of SURVEY.md Appendix C, `kernel` in TF
layout [Din+H, 4H] with gate blocks i, ci, f, o).  Not kept: the TensorBundle container -- the payload
is one .npz per checkpoint holding {variable name: float32 array} plus the optimizer slots, so a reader
needs numpy only.  `sess` is accepted and ignored (there is no session).
"""
import os
import re

import numpy as np
import torch

_OPT = '__optimizer__/'


class CheckpointState(object):
    def __init__(self, model_checkpoint_path, all_model_checkpoint_paths):
        self.model_checkpoint_path = model_checkpoint_path
        self.all_model_checkpoint_paths = list(all_model_checkpoint_paths)


def _index_path(directory):
    return os.path.join(directory, 'checkpoint')


def get_checkpoint_state(checkpoint_dir):
    """tf.train.get_checkpoint_state: None when the directory holds no `checkpoint` index."""
    idx = _index_path(checkpoint_dir)
    if not os.path.isfile(idx):
        return None
    latest, allp = None, []
    for line in open(idx):
        m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$', line)
        if not m:
            continue
        path = m.group(2)
        if not os.path.isabs(path):
            path = os.path.join(checkpoint_dir, path)
        if m.group(1) == 'model_checkpoint_path':
            latest = path
        else:
            allp.append(path)
    if latest is None:
        return None
    return CheckpointState(latest, allp or [latest])


class Saver(object):
    def __init__(self, max_to_keep=None):
        self.max_to_keep = max_to_keep
        self._kept = []

    @staticmethod
    def _file(prefix):
        return prefix + '.npz'

    def save(self, sess, save_path, global_step=None, model=None):
        """Writes `<save_path>-<global_step>.npz` and updates the directory's `checkpoint` index.
        `sess` may be the model itself (there is no session); returns the checkpoint prefix."""
        model = model if model is not None else sess
        prefix = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
        if model.store.flat.is_cuda:
            # a checkpoint is a sync point anyway: never persist weights trained through a timed-out hand-off
            from ... import ops
            ops.check_async_errors(model.store.flat.device.index or 0)
        arrays = {n: v.detach().cpu().numpy() for n, v in model.store.state_dict().items()}
        opt = getattr(model, 'optimizer', None)
        if opt is not None:
            arrays[_OPT + 'name'] = np.array(opt.name)
            arrays[_OPT + 'global_step'] = np.array(opt.global_step, dtype=np.int64)
            for k in ('slot0', 'slot1'):
                t = getattr(opt, k)
                if t is not None:
                    arrays[_OPT + k] = t.detach().cpu().numpy()
        directory = os.path.dirname(prefix) or '.'
        os.makedirs(directory, exist_ok=True)
        tmp = self._file(prefix) + '.tmp'
        with open(tmp, 'wb') as f:
            np.savez(f, **arrays)
        os.replace(tmp, self._file(prefix))
        if prefix in self._kept:
            self._kept.remove(prefix)
        self._kept.append(prefix)
        if self.max_to_keep:
            while len(self._kept) > self.max_to_keep:
                old = self._kept.pop(0)
                if os.path.isfile(self._file(old)):
                    os.remove(self._file(old))
        with open(_index_path(directory), 'w') as f:
            f.write('model_checkpoint_path: "%s"\n' % os.path.basename(prefix))
            for p in self._kept:
                f.write('all_model_checkpoint_paths: "%s"\n' % os.path.basename(p))
        return prefix

    def restore(self, sess, save_path, model=None):
        """Loads the variables and the optimizer state (slots, global_step); an optimizer the model has not created yet
        is built from the checkpoint's record."""
        model = model if model is not None else sess
        path = self._file(save_path)
        if not os.path.isfile(path):
            raise ValueError('checkpoint %s does not exist' % path)
        with np.load(path, allow_pickle=False) as z:
            names = model.store.names
            missing = [n for n in names if n not in z.files]
            if missing:
                raise ValueError('checkpoint %s lacks variables: %s' % (path, ', '.join(missing[:5])))
            model.store.load_state_dict({n: torch.from_numpy(np.asarray(z[n], dtype=np.float32)) for n in names})
            opt = getattr(model, 'optimizer', None)
            if (_OPT + 'name') in z.files:
                saved = str(z[_OPT + 'name'])
                if opt is None and hasattr(model, '_set_optimizer'):
                    # the recipes create the optimizer lazily inside train(): build it now so that a
                    # restore-then-train resume continues the saved slots / Adam step (train() adopts it,
                    # only the learning rate is replaced)
                    opt = model.optimizer = model._set_optimizer(saved, 0.0)
                if opt is not None and saved == opt.name:
                    opt.global_step = int(z[_OPT + 'global_step'])
                    for k in ('slot0', 'slot1'):
                        t = getattr(opt, k)
                        if t is not None and (_OPT + k) in z.files:
                            t.copy_(torch.from_numpy(z[_OPT + k]).to(t.device))
                elif opt is not None:
                    import warnings
                    warnings.warn('checkpoint %s holds %s optimizer state but the model uses %s: slots and '
                                  'global_step restart from their initial values' % (path, saved, opt.name))
        return model


def sync_point():
    """End of an epoch / of training: every asynchronously watched condition of the steps issued so far is inspected NOW
    (ops.flush_deferred_checks: the "labels do not fit the frames" error of tf.nn.ctc_loss, which the reference raises
    inside the offending sess.run and this backend otherwise reports up to ops.DeferredCheck.DEPTH steps late).  Saver.save
    does the same through ops.check_async_errors.  A no-op without pending checks."""
    from ... import ops
    flush = getattr(ops, 'flush_deferred_checks', None)
    if flush is not None:
        flush()

