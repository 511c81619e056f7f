"""ctypes binding of libppyolo_hip.so (the C ABI declared in include/ppyolo_hip.h).

`cffi` is not installed in this image, so the thin C-ABI layer is
bound with the stdlib.  There is NO fallback: if the shared library is missing the
import of any op raises, and every call checks the C return code.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PPYOLO_HIP_LIB') or os.path.join(_HERE, 'lib', 'libppyolo_hip.so')     # (override: experiments)

OK = 0
ACT = {None: 0, 'relu': 1, 'leaky': 2}


class PPYoloHipError(RuntimeError):
    pass


_lib = None

_PROTOS = {
    'ppy_version': (c_int, []),
    'ppy_error_string': (ctypes.c_char_p, [c_int]),
    'ppy_last_hip_error': (ctypes.c_char_p, []),
    'ppy_note_hip_error': (None, [c_int]),
    'ppy_conv2d_split_weights_bf16x3': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    'ppy_conv2d_split_weights_f16x2': (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ppy_conv2d_bn_act_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 13
                              + [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_conv2d_bn_act_split_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 13
                                    + [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p]),
    'ppy_conv2d_workspace_bytes': (c_size_t, [c_int] * 11),
    'ppy_conv2d_bn_partials_bytes': (c_size_t, [ctypes.c_longlong, c_int]),
    'ppy_conv2d_train_fwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 10
                                 + [c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_int), c_void_p]),
    'ppy_bn_train_stats_merge_f32': (c_int, [c_void_p, c_size_t, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ppy_conv1x1_expand_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]
                               + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]),
    'ppy_conv3x3_maxpool_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 6 + [c_void_p, c_void_p, c_void_p]),
    'ppy_conv1x1_stats_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    'ppy_conv1x1_bn_apply_f32': (c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_void_p, c_int] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]),
    'ppy_conv2d_num_configs': (c_int, []),
    'ppy_conv2d_stream_first_config': (c_int, []),
    'ppy_conv2d_patch_first_config': (c_int, []),
    'ppy_conv2d_ws_first_config': (c_int, []),
    'ppy_conv2d_small_first_config': (c_int, []),
    'ppy_conv2d_pick': (c_int, [c_int] * 9 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'ppy_conv2d_dgrad_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int] + [c_int] * 11 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_conv2d_dgrad_workspace_bytes': (c_size_t, [c_int] * 11),
    'ppy_train_prepare_weights_f16x2': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'ppy_conv2d_dgrad_prepared_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int] + [c_int] * 10
                                      + [c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_conv2d_wgrad_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_conv2d_wgrad_workspace_bytes': (c_size_t, [c_int] * 9),
    'ppy_bn_train_workspace_bytes': (c_size_t, [c_int, c_int]),
    'ppy_bn_train_stats_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    'ppy_bn_train_apply_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_void_p, c_void_p]),
    'ppy_bn_train_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_act_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_longlong, c_int, c_int, c_void_p]),
    'ppy_upsample2x_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ppy_spp_bwd_workspace_bytes': (c_size_t, [c_int] * 4),
    'ppy_spp_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    'ppy_dropblock_workspace_bytes': (c_size_t, [c_int] * 4),
    'ppy_dropblock_mask_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_ulonglong, c_void_p, c_size_t,
                                        c_void_p]),
    'ppy_dropblock_apply_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_int, c_void_p]),
    'ppy_sgd_momentum_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_int, c_void_p]),
    'ppy_avgpool2x2_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 4 + [c_void_p]),
    'ppy_maxpool3x3s2_bwd_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int] + [c_int] * 4 + [c_void_p]),
    'ppy_zero_insert_f32': (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 7 + [c_void_p]),
    'ppy_ema_update_f32': (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_float, c_void_p]),
    'ppy_add_inplace_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_longlong, c_int, c_void_p]),
    'ppy_upsample2x_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ppy_channel_sum_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_yolov3_loss_workspace_bytes': (c_size_t, [c_int] * 3),
    'ppy_yolov3_loss_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_float), c_int, c_int, c_int, c_int, c_int,
                                     c_double, c_double, c_double, c_int, c_int, c_double, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    'ppy_stem_conv3x3s2_nchw_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                             c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ppy_stem_conv3x3s2_nchw_x3_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ppy_preprocess_u8_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    'ppy_maxpool3x3s2_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ppy_avgpool2x2_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ppy_spp_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                            c_void_p]),
    'ppy_dcnv2_sample_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 8 + [c_void_p]),
    'ppy_dcnv2_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_int] + [c_int] * 10 + [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ppy_dcnv2_workspace_bytes': (c_size_t, [c_int] * 9),
    'ppy_dcnv2_num_configs': (c_int, []),
    'ppy_dcnv2_backward_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]
                               + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    'ppy_dcnv2_backward_workspace_bytes': (c_size_t, [c_int] * 7),
    'ppy_yolo_decode_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_float), c_int,
                                    c_double, c_int, c_double, c_int, c_void_p, c_void_p, c_int, c_int, c_float,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'ppy_yolo_decode_levels_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                           c_int, c_double, c_int, c_double, c_int, c_void_p, c_void_p, c_int, c_float,
                                           c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'ppy_matrix_nms_f32': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                   c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    'ppy_matrix_nms_workspace_bytes': (c_size_t, [c_int]),
    'ppy_nms_candidates_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p]),
    'ppy_conv3x3_conv1x1_f32': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_int, c_void_p, c_int] + [c_int] * 6 + [c_float, c_float, c_void_p, c_void_p]),
    'ppy_lane_stream_create': (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_uint32), c_int]),
    'ppy_lane_stream_destroy': (c_int, [c_void_p]),
}


def exported_symbols():
    """Every entry point include/ppyolo_hip.h declares."""
    return sorted(_PROTOS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PPYoloHipError(
                'libppyolo_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` (hipcc, gfx950). There is no CPU / PyTorch fallback for the HIP path.' % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64; the process must hold ONE HIP runtime, the one torch initialises
        # (device memory and streams come from torch).  Loading this library first binds it to the system runtime
        # instead, and its launches then fail with hipErrorNoDevice -- so make sure torch's is mapped before dlopen.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != OK:
        msg = lib().ppy_error_string(rc).decode()
        if rc == -4:                  # PPY_ERR_LAUNCH: say which HIP error it was
            msg += ' [%s]' % lib().ppy_last_hip_error().decode()
        raise PPYoloHipError('%s failed: %s (code %d)' % (what or 'libppyolo_hip call', msg, rc))
