"""ctypes binding of libasr_hip.so (include/asr_hip.h).

There is NO fallback: if the shared library is missing or a call fails this
module raises.  torch is used only to own device memory / streams.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libasr_hip.so')

ASR_F32, ASR_BF16 = 0, 1
OPTIMIZER_IDS = {'sgd': 0, 'momentum': 1, 'nestrov': 2, 'adagrad': 3, 'adadelta': 4,
                 'rmsprop': 5, 'adam': 6}

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_i64, _u64 = C.c_int64, C.c_uint64

# name -> (restype, argtypes); mirrors include/asr_hip.h declaration by declaration
SIGNATURES = {
    'asr_abi_version': (_i, []),
    'asr_create': (_i, [C.POINTER(_vp), _i]),
    'asr_create_ex': (_i, [C.POINTER(_vp), _i, _sz]),
    'asr_scratch_bytes': (_sz, [_vp]),
    'asr_destroy': (_i, [_vp]),
    'asr_set_xcd_skip': (_i, [_vp, _i]),
    'asr_set_gemm_tn_workgroups': (_i, [_vp, _i]),
    'asr_last_error_string': (C.c_char_p, [_vp]),
    'asr_device_info': (_i, [_vp, C.POINTER(_i), C.c_char_p, _i]),
    'asr_bt_to_tb': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    'asr_bt_to_tb_ld': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'asr_transpose2d': (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    'asr_cast_from_f32': (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    'asr_cast_to_f32': (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    'asr_apply_mask': (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'asr_dropout_mask': (_i, [_vp, _vp, _sz, _f, _u64, _u64, _vp]),
    'asr_touch': (_i, [_vp, _vp, _sz, _vp]),
    'asr_dropout_apply': (_i, [_vp, _i, _vp, _vp, _sz, _f, _u64, _u64, _vp]),
    'asr_relu_bwd_drop': (_i, [_vp, _i, _vp, _vp, _sz, _f, _u64, _u64, _vp, _vp]),
    'asr_colsum': (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    'asr_gemm': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'asr_gemm_act': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    'asr_gemm_drop': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _f, _u64, _u64, _vp]),
    'asr_gemm_mul': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    'asr_conv3x3_prep_weights': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    'asr_conv3x3_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'asr_conv3x3_fwd_drop': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _u64, _u64, _vp, _vp]),
    'asr_conv3x3_bwd_data': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    'asr_conv3x3_smallc_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'asr_conv3x3_smallc_fwd_drop': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _f, _u64, _u64, _vp, _vp]),
    'asr_conv3x3_smallc_bwd_weight': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_conv3x3_smallc_bwd_weight_bias': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'asr_conv3x3_bwd_data_relu': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _f, _u64, _u64, _i, _vp, _vp]),
    'asr_conv3x3_bwd_weight': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'asr_conv3x3_bwd_weight_bias': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'asr_im2col3x3': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_col2im3x3': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_im2col': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_col2im': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_maxpool2x2_fwd': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'asr_maxpool2x2_fwd_drop': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _u64, _u64, _vp]),
    'asr_relu_bwd_scaled': (_i, [_vp, _i, _vp, _vp, _sz, _f, _vp, _vp]),
    'asr_maxpool2x2_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'asr_relu_bwd': (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    'asr_maxpool2x2_relu_bwd': (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _f, _u64, _u64, _i, _vp]),
    'asr_lstm_prep_weights': (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'asr_lstm_prep_layer': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'asr_lstm_grad_finish': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    'asr_gate_deinterleave': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    'asr_lstm_fwd': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'asr_lstm_bwd': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp]),
    'asr_lstm_bwd_ex': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp,
                             _vp, _vp]),
    'asr_gru_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'asr_gru_bwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'asr_check_async_errors': (_i, [_vp, C.POINTER(C.c_uint)]),
    'asr_peek_async_errors': (_i, [_vp, _vp, _vp]),
    'asr_clear_async_errors': (_i, [_vp, _vp]),
    'asr_debug_set_lstm_flags': (_i, [_i]),
    'asr_debug_set_gru_persistent': (_i, [_i]),
    'asr_debug_placement': (_i, [_vp, _vp, _i, _i, _vp]),
    'asr_debug_poison_lds': (_i, [_vp, _vp]),
    'asr_debug_tear_probe': (_i, [_vp, _vp, C.c_uint, _i, _i, _vp, _vp]),
    'asr_ctc_workspace_bytes': (_sz, [_i, _i, _i]),
    'asr_ctc_loss': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'asr_ctc_greedy_decode': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'asr_ctc_beam_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'asr_ctc_beam_decode': (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    'asr_softmax_rows': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'asr_lstm_cell_fwd': (_i, [_vp] * 6 + [_i, _i, _f, _f] + [_vp] * 6),
    'asr_lstm_cell_fwd_ex': (_i, [_vp] * 6 + [_i, _i, _f, _f] + [_vp] * 5 + [_vp, _vp, _vp, _i, _vp, _i, _vp]),
    'asr_lstm_cell_gemm_ok': (_i, [_i, _i, _i, _i]),
    'asr_lstm_cell_gemm_prep': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    'asr_lstm_cell_gemm_fwd': (_i, [_vp, _vp, _i, _i, _vp, _i] + [_vp] * 4 + [_i, _i, _f, _f] + [_vp] * 5 +
                               [_vp, _vp, _vp, _i, _vp, _i, _vp]),
    'asr_lstm_cell_gemm_h_bytes': (_sz, [_i, _i]),
    'asr_lstm_cell_gemm_prep_h': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    'asr_lstm_cell_gemm_fwd_h': (_i, [_vp, _vp, _i, _i, _vp] + [_vp] * 4 + [_i, _i, _f, _f] + [_vp] * 5 +
                                 [_vp, _vp, _vp, _i, _vp, _i, _vp]),
    'asr_lstm_cell_gemm_bwd_h': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    'asr_lstm_cell_bwd': (_i, [_vp] * 9 + [_i, _i] + [_vp] * 5),
    'asr_lstm_cell_bwd_ex': (_i, [_vp] * 9 + [_i, _i, _f] + [_vp] * 5),
    'asr_stack_frames': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'asr_splice': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'asr_att_energy_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'asr_att_energy_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'asr_att_softmax_ctx_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'asr_att_softmax_ctx_fwd_ex': (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    'asr_att_softmax_ctx_bwd': (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'asr_att_loc_energy_fwd': (_i, [_vp] * 7 + [_i, _i, _i, _i, _vp, _vp]),
    'asr_att_loc_energy_bwd': (_i, [_vp] * 8 + [_i, _i, _i, _i] + [_vp] * 6 + [_i, _vp]),
    'asr_att_decoder_fwd': (_i, [_vp, _vp, _vp]),
    'asr_att_decoder_bwd': (_i, [_vp, _vp, _vp]),
    'asr_att_decoder_infer': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'asr_add_cols': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    'asr_tanh_fwd': (_i, [_vp, _vp, _vp, _sz, _vp]),
    'asr_tanh_bwd': (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    'asr_embedding_gather': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    'asr_embedding_scatter': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'asr_seq_xent': (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp]),
    'asr_argmax_rows': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'asr_clip_plan': (_i, [_vp, _vp, _i, _vp]),
    'asr_clip_by_norm_multi': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _f, _vp, _vp]),
    'asr_weight_decay': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    'asr_optimizer_step': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _f, _i64, _vp]),
    'asr_scale': (_i, [_vp, _vp, _sz, _f, _vp]),
    'asr_comm_set_library': (_i, [C.c_char_p]),
    'asr_comm_unique_id': (_i, [_vp]),
    'asr_comm_init': (_i, [C.POINTER(_vp), _vp, _i, _i, _vp]),
    'asr_comm_destroy': (_i, [_vp]),
    'asr_comm_info': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'asr_allreduce_mean': (_i, [_vp, _vp, _sz, _vp]),
}

_lib = None


ABI_VERSION = 5          # include/asr_hip.h: asr_abi_version()


def load():
    """dlopen libasr_hip.so and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libasr_hip.so not found at %s -- build it with '
            '`python -m tensorflow_end2end_speech_recognition_amd.build` '
            '(there is no CPU fallback for the HIP path)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    ver = getattr(lib, 'asr_abi_version', None)
    got = int(ver()) if ver is not None else 0
    if got < ABI_VERSION:
        raise RuntimeError('%s reports asr_abi_version() = %d but this package binds version %d entry points: the '
                           'library is stale -- rebuild it with `python -m tensorflow_end2end_speech_recognition_amd.build`'
                           % (LIB_PATH, got, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError('%s (abi %d) does not export %s: rebuild it with '
                               '`python -m tensorflow_end2end_speech_recognition_amd.build`' % (LIB_PATH, got, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class AsrError(RuntimeError):
    pass


class Handle(object):
    """One per process per GPU (asr_create/asr_destroy)."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.asr_create(C.byref(h), int(device))
        if rc != 0:
            raise AsrError('asr_create(device=%d) failed with %d: no usable MI355X/HIP device -- '
                           'the HIP path has no CPU fallback' % (device, rc))
        self.h = h
        self.device = device

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.asr_last_error_string(self.h)
            msg = msg.decode() if msg else ''
            if rc == -1:
                raise ValueError('%s: %s' % (what, msg))
            raise AsrError('%s failed (%d): %s' % (what, rc, msg))

    def info(self):
        n = _i(0)
        buf = C.create_string_buffer(128)
        self.check(self.lib.asr_device_info(self.h, C.byref(n), buf, 128), 'asr_device_info')
        return n.value, buf.value.decode()

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.asr_destroy(self.h)
                self.h = None
        except Exception:
            pass


_handles = {}


def handle(device=0, lane=0):
    """lane 0 = the main launch stream's handle; lane 1 = a second handle (own split-K scratch)
    for work issued on the side stream (ops.side_lane), so concurrent GEMMs never share scratch."""
    key = (device, lane)
    if key not in _handles:
        _handles[key] = Handle(device)
    return _handles[key]
