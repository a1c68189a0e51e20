"""Base class for all models -- mirror of models/model_base.py:23-166.

The reference builds TF graph ops; here every method executes eagerly on the
HIP kernels.  Kept: OPTIMIZER_CLS_NAMES keys (model_base.py:12-20, including the
'nestrov' spelling), _set_optimizer (momentum 0.9 for momentum/nestrov),
train(loss, optimizer, learning_rate), _clip_gradients (per-variable
tf.clip_by_norm, model_base.py:148-152) and the ValueError on unknown names.
"""
import torch

from .. import ops
from .._lib import OPTIMIZER_IDS

OPTIMIZER_CLS_NAMES = dict((k, k) for k in OPTIMIZER_IDS)

# TF1 slot initial values: Adagrad accumulator 0.1, RMSProp rms ones
_SLOT0_INIT = {'adagrad': 0.1, 'rmsprop': 1.0}


class Optimizer(object):
    """tf.train.*Optimizer stand-in over the model's flat parameter buffer."""

    def __init__(self, name, learning_rate, store):
        self.name = name
        self.opt_id = OPTIMIZER_IDS[name]
        self.learning_rate = learning_rate
        self.store = store
        self.slot0 = self.slot1 = None
        if name != 'sgd':
            self.slot0 = torch.full_like(store.flat, _SLOT0_INIT.get(name, 0.0))
        if name in ('adadelta', 'rmsprop', 'adam'):
            self.slot1 = torch.zeros_like(store.flat)
        self.global_step = 0

    def compute_gradients(self, loss, model=None):
        """Runs the backward pass of the model that produced `loss`; returns the list of
        (gradient view, variable name) pairs (train_ctc.py:112 tower_grads)."""
        model = model or getattr(loss, '_asr_model', None)
        if model is None:
            raise ValueError('compute_gradients needs the model that produced the loss')
        model._backward()
        st = self.store
        return [(st.g(n), n) for n in st.names]

    def apply_gradients(self, grads_and_vars=None, global_step=None, learning_rate=None):
        lr = self.learning_rate if learning_rate is None else learning_rate
        self.global_step += 1
        st = self.store
        ops.optimizer_step(self.opt_id, st.flat, st.grad, self.slot0, self.slot1, lr, self.global_step)
        st.mark_dirty()
        if st.flat.is_cuda:
            # hand-off timeouts of the multi-CU recurrence kernels must not train on silently (raises AsrError)
            ops.watch_async_errors(st.flat.device)
        return self.global_step

    def state_dict(self):
        return {'slot0': self.slot0, 'slot1': self.slot1, 'global_step': self.global_step,
                'name': self.name}


class ModelBase(object):

    def __init__(self, *args, **kwargs):
        self.optimizer = None
        self.clip_grad_norm = None
        self.store = None

    def _build(self, *args, **kwargs):
        """Construct model graph."""
        raise NotImplementedError  # (the reference raises NotADirectoryError by typo, model_base.py:30)

    def create_placeholders(self):
        """Create placeholders and append them to list."""
        raise NotImplementedError

    def compute_loss(self, *args, **kwargs):
        """Operation for computing loss."""
        raise NotImplementedError

    def _add_noise_to_inputs(self, inputs, stddev=0.075):
        raise NotImplementedError

    def _add_noise_to_gradients(self, grads_and_vars, gradient_noise_scale, stddev=0.075):
        raise NotImplementedError

    def _set_optimizer(self, optimizer, learning_rate):
        """model_base.py:68-95."""
        optimizer = optimizer.lower()
        if optimizer not in OPTIMIZER_CLS_NAMES:
            raise ValueError(
                "Optimizer name should be one of [%s], you provided %s." %
                (", ".join(OPTIMIZER_CLS_NAMES), optimizer))
        return Optimizer(optimizer, learning_rate, self.store)

    def train(self, loss, optimizer, learning_rate):
        """One training step for the forward pass that produced `loss`
        (model_base.py:97-133: compute_gradients -> _clip_gradients -> apply_gradients)."""
        if self.optimizer is None or self.optimizer.name != optimizer.lower():
            self.optimizer = self._set_optimizer(optimizer, learning_rate)
        self.optimizer.learning_rate = learning_rate
        grads_and_vars = self.optimizer.compute_gradients(loss, model=self)
        if self.clip_grad_norm is not None:
            grads_and_vars = self._clip_gradients(grads_and_vars)
        return self.optimizer.apply_gradients(grads_and_vars)

    def _clip_gradients(self, grads_and_vars):
        """Per-variable tf.clip_by_norm(grad, clip_norm=self.clip_grad_norm), model_base.py:135-166,
        as one multi-tensor pass over the flat gradient buffer."""
        ops.clip_by_norm_multi(self.store.grad, self.store.plan, float(self.clip_grad_norm))
        return grads_and_vars
