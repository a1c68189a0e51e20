import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _bounded_cpu_threads():
    """The oracle's per-time-step loops do thousands of tiny matmuls: on a 256-core GPU-box host an unbounded
    BLAS / torch thread pool spends its time waking threads (measured: 9.5 s vs 0.6 s for one 300-step layer)."""
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    try:
        from threadpoolctl import threadpool_limits
        ctl = threadpool_limits(limits=4, user_api='blas')
    except Exception:       # threadpoolctl absent: run unbounded
        ctl = None
    yield
    if ctl is not None:
        ctl.restore_original_limits()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


_PEEK = {}


@pytest.fixture(autouse=True)
def _gpu_test_aids(request):
    """ASR_POISON_LDS=1 (GPU box): every CU's LDS holds NaN patterns when a GPU test starts (asr_debug_poison_lds), so a
    kernel that reads LDS words nobody wrote fails every time instead of once in a few cold starts.  Together with
    ASR_POISON_SCRATCH=1 (the handle's work arena) and scripts/poison_pytest.py (torch's caching allocator)."""
    if os.environ.get('ASR_POISON_LDS') == '1' and request.node.get_closest_marker('gpu') is not None:
        import torch
        if torch.cuda.is_available():
            from tensorflow_end2end_speech_recognition_amd import _lib, ops
            h = _lib.handle(0, 0)
            h.check(h.lib.asr_debug_poison_lds(h.h, ops._s()), 'asr_debug_poison_lds')
    yield
    # ASR_PEEK_STICKY=<file>: the same question without draining the device after every test (a drain hides anything
    # that depends on the previous test's last launches still running): an asynchronous copy of the word behind each
    # test, looked at one test later
    plog = os.environ.get('ASR_PEEK_STICKY')
    if plog and request.node.get_closest_marker('gpu') is not None:
        import torch
        if torch.cuda.is_available():
            from tensorflow_end2end_speech_recognition_amd import _lib, ops
            import ctypes
            st = _PEEK
            if st.get('host') is None:
                st['host'] = torch.zeros(2, dtype=torch.int32).pin_memory()
                st['n'] = 0
                st['pending'] = None
            if st['pending'] is not None:
                name, slot, ev = st['pending']
                ev.synchronize()
                v = int(st['host'][slot])
                if v:
                    with open(plog, 'a') as f:
                        f.write('%s: word 0x%x behind it\n' % (name, v))
            slot = st['n'] % 2
            h = _lib.handle(0, 0)
            h.lib.asr_peek_async_errors(h.h, ctypes.c_void_p(st['host'].data_ptr() + 4 * slot), ops._s())
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            st['pending'] = (request.node.nodeid, slot, ev)
            st['n'] += 1
    # ASR_CHECK_STICKY=<file>: which test leaves the cluster kernels' sticky error word set behind it?  (appends its id)
    log = os.environ.get('ASR_CHECK_STICKY')
    if log and request.node.get_closest_marker('gpu') is not None:
        import ctypes
        import torch
        if torch.cuda.is_available():
            from tensorflow_end2end_speech_recognition_amd import _lib, ops
            h = _lib.handle(0, 0)
            flags = ctypes.c_uint(0)
            h.lib.asr_check_async_errors(h.h, ctypes.byref(flags))
            if flags.value:
                h.lib.asr_clear_async_errors(h.h, ops._s())
                with open(log, 'a') as f:
                    f.write('%s left flags 0x%x\n' % (request.node.nodeid, flags.value))
