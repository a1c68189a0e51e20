"""CPU: the C-ABI library builds, loads, and exports every symbol include/asr_hip.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'asr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(asr_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_are_exported_and_typed():
    from tensorflow_end2end_speech_recognition_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from tensorflow_end2end_speech_recognition_amd.build import build
        build(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'libasr_hip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'ctypes binding misses %s' % n
    assert set(_lib.SIGNATURES) <= set(names), set(_lib.SIGNATURES) - set(names)
    assert lib.asr_abi_version() == _lib.ABI_VERSION == 5


def test_no_cpu_fallback():
    import torch
    from tensorflow_end2end_speech_recognition_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.AsrError):
        _lib.Handle(0)
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_host_label_formats():
    import numpy as np
    from tensorflow_end2end_speech_recognition_amd.utils.io.labels.sparsetensor import (
        dense_to_flat, list2sparsetensor, sparse_to_flat, sparsetensor2list)
    dense = np.array([[1, 2, -1, -1], [3, -1, -1, -1], [4, 5, 6, 7]])
    st = list2sparsetensor(dense, -1)
    assert st[0].tolist() == [[0, 0], [0, 1], [1, 0], [2, 0], [2, 1], [2, 2], [2, 3]]
    assert st[2].tolist() == [3, 4]
    flat, off, mx = sparse_to_flat(st, 3)
    f2, o2, m2 = dense_to_flat(dense, -1)
    assert flat.tolist() == f2.tolist() == [1, 2, 3, 4, 5, 6, 7]
    assert off.tolist() == o2.tolist() == [0, 2, 3, 7] and mx == m2 == 4
    back = sparsetensor2list(st, 3)
    assert [list(b) for b in back] == [[1, 2], [3], [4, 5, 6, 7]]


def test_registry_and_errors():
    from tensorflow_end2end_speech_recognition_amd.models.encoders.load_encoder import load
    from tensorflow_end2end_speech_recognition_amd.models.model_base import OPTIMIZER_CLS_NAMES
    assert load('blstm').__name__ == 'BLSTMEncoder' and load('lstm').__name__ == 'LSTMEncoder'
    with pytest.raises(ValueError):
        load('no_such_encoder')
    assert set(OPTIMIZER_CLS_NAMES) == {'adagrad', 'adadelta', 'adam', 'rmsprop', 'sgd', 'momentum', 'nestrov'}


def test_multitask_model_construction_contract():
    """MultitaskCTC / multitask encoders (models/ctc/multitask_ctc.py:62-98, multitask_blstm.py:66-68): variable
    names and creation order, argument validation, the unidirectional encoder's list-alias quirk.  (Host logic only:
    the parameter store lives on the CPU here; compute is covered by the GPU test.)"""
    from tensorflow_end2end_speech_recognition_amd.models.encoders.load_encoder import load
    from tensorflow_end2end_speech_recognition_amd.models.ctc.multitask_ctc import MultitaskCTC
    assert load('multitask_blstm').__name__ == 'MultitaskBLSTMEncoder'
    assert load('multitask_lstm').__name__ == 'MultitaskLSTMEncoder'
    m = MultitaskCTC('multitask_blstm', 12, 64, 3, 2, 7, 4, 0.7, bottleneck_dim=16, device='cpu')
    names = list(m.store.state_dict().keys())
    assert names[-6:] == ['output_sub/weights', 'output_sub/biases', 'bottleneck/weights', 'bottleneck/biases',
                          'output_main/weights', 'output_main/biases']
    assert m.num_classes == 8 and m.num_classes_sub == 5 and abs(m.sub_task_weight - 0.3) < 1e-12
    assert m.encoder.num_layers == 3 and m.encoder.num_layers_sub == 2 and m.name == 'multitask_blstm_ctc'
    assert MultitaskCTC('multitask_lstm', 12, 64, 3, 1, 7, 4, 0.5, device='cpu').encoder.num_layers_sub == 3
    with pytest.raises(ValueError):
        MultitaskCTC('multitask_blstm', 12, 64, 2, 3, 7, 4, 0.5, device='cpu')      # sub deeper than main
    with pytest.raises(ValueError):
        MultitaskCTC('multitask_blstm', 12, 64, 2, 1, 7, 4, 1.5, device='cpu')      # weight outside [0, 1]
    with pytest.raises(NotImplementedError):
        MultitaskCTC('blstm', 12, 64, 2, 1, 7, 4, 0.5, device='cpu')
