"""Oracle (test infrastructure; update rules pinned to TensorFlow's own optimizer tests where those hold numbers --
tests/golden/tf_known_answers.py: sgd, momentum, nesterov, adagrad, adadelta, rmsprop, adam; clip_by_norm per variable
also to the reference's ModelBase._clip_gradients as executed, tests/test_oracle_tfshim.py):
numpy restatement of
tf.clip_by_norm and the seven tf.train optimizers the reference selects in
models/model_base.py:12-20,68-95 with TF1 default hyper-parameters
(SURVEY.md Appendix B), plus the tower mean of utils/training/multi_gpu.py:13-48."""
import numpy as np


def clip_by_norm(g, clip):
    """t * clip / max(||t||_2, clip)  (models/model_base.py:151)."""
    n = np.sqrt((g.astype(np.float64) ** 2).sum())
    return g * (clip / max(n, clip))


def init_slots(name, p):
    s0 = np.zeros_like(p)
    s1 = np.zeros_like(p)
    if name == 'adagrad':
        s0[:] = 0.1
    if name == 'rmsprop':
        s0[:] = 1.0
    return s0, s1


def step(name, p, g, s0, s1, lr, t, momentum=0.9, decay=0.9, rms_eps=1e-10):
    """One update; returns (p, s0, s1).  t is the 1-based step count.  momentum / decay / rms_eps default to what the
    reference passes or leaves at TensorFlow's defaults (model_base.py:82-95); TensorFlow's own optimizer tests, which
    tests/test_oracle.py pins the rules to, use other values for some of them."""
    if name == 'sgd':
        p = p - lr * g
    elif name == 'momentum':
        s0 = momentum * s0 + g
        p = p - lr * s0
    elif name == 'nestrov':
        s0 = momentum * s0 + g
        p = p - (lr * g + lr * momentum * s0)
    elif name == 'adagrad':
        s0 = s0 + g * g
        p = p - lr * g / np.sqrt(s0)
    elif name == 'adadelta':
        rho, eps = 0.95, 1e-8
        s0 = rho * s0 + (1 - rho) * g * g
        upd = np.sqrt(s1 + eps) / np.sqrt(s0 + eps) * g
        s1 = rho * s1 + (1 - rho) * upd * upd
        p = p - lr * upd
    elif name == 'rmsprop':
        s0 = decay * s0 + (1 - decay) * g * g
        s1 = lr * g / np.sqrt(s0 + rms_eps)
        p = p - s1
    elif name == 'adam':
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        s0 = b1 * s0 + (1 - b1) * g
        s1 = b2 * s1 + (1 - b2) * g * g
        p = p - lr_t * s0 / (np.sqrt(s1) + eps)
    else:
        raise ValueError(name)
    return p, s0, s1


def average_gradients(tower_grads):
    """utils/training/multi_gpu.py:13-48: mean over towers, per variable."""
    return [np.mean(np.stack(gs, 0), 0) for gs in zip(*tower_grads)]
