"""Bitwise run-to-run determinism of the training step, across FRESH processes.

Every key of tests/_determinism_worker.py (cfg A 2x128 fp32, the 5x256 bf16 headline model, a VGG-BLSTM, a Bahdanau
attention model, a cfg-D-shaped joint location-attention model with carried weights) runs two consecutive training steps
(+ greedy inference for the attention models) in N fresh processes with the handle's work arena and every CU's LDS
poisoned (ASR_POISON_SCRATCH / ASR_POISON_LDS) -- the first of them is the first process to touch the device after this
test module started.  All digests (loss, logits, the whole gradient, the updated parameters; raw bytes) must be equal in
every process: split-K slabs are reduced in a fixed order, the CTC occupation sums and every colsum run in a fixed order,
nothing accumulates with atomics, dropout is a counter-based stream -- so a difference is a race or a read of memory
nobody wrote, not rounding."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PROC = int(os.environ.get('ASR_DETERMINISM_PROCS', '5'))


def _run(key, extra_env=None):
    env = dict(os.environ, ASR_POISON_SCRATCH='1', ASR_POISON_LDS='1', PYTHONPATH=ROOT)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_determinism_worker.py'), key], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (key, r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('DIGEST ')][-1]
    return json.loads(line[len('DIGEST '):])


@pytest.mark.parametrize('key', ['bahdanau', 'cfgA', 'headline', 'vgg', 'cfgD_toy'])
def test_training_step_is_bitwise_reproducible_across_fresh_processes(cuda, key):
    runs = [_run(key) for _ in range(N_PROC)]
    first = runs[0]
    for i, r in enumerate(runs[1:], 1):
        assert r == first, 'process %d of %s differs from process 0:\n%s\n%s' % (
            i, key, json.dumps(first, sort_keys=True), json.dumps(r, sort_keys=True))
    assert first['step1'] != first['step2']                  # the update moved the parameters


def test_exchange_flavours_give_the_same_bits(cuda):
    """The recurrence's cluster hand-off in its write-through / placement-independent form (ASR_LSTM_DFLAGS=16) must not
    change a bit of the headline model's step."""
    a = _run('headline')
    b = _run('headline', {'ASR_LSTM_DFLAGS': '16'})
    assert a == b
