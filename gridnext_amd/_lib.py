"""ctypes binding of libgridnext_hip.so (the C-ABI declared in include/gridnext_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor
is not on a HIP device, the call raises.  The product path never computes on the CPU.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GNX_LIB') or os.path.join(_HERE, 'libgridnext_hip.so')   # GNX_LIB: debug builds only

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes)   [stream is always the last pointer]
SIGNATURES = {
    'gnx_hexconv_fwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_hexconv_bwd_data': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_hexconv_bwd_weight_workspace': (_L, [_I, _I, _I, _I, _I]),
    'gnx_hexconv_bwd_weight': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_hexconv_bwd_weight_batch': (_I, [_P, _I, _P]),
    'gnx_bn_workspace': (_L, [_L, _I]),
    'gnx_bn_train_stats': (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    'gnx_bn_train_stats_sync': (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    'gnx_bn_train_stats_apply': (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    'gnx_bn_sync_words': (_L, [_I]),
    'gnx_bn_train_stats_apply_sync': (_I, [_P, _L, _L, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    'gnx_bn_relu_bwd_sync': (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'gnx_bn_fold_eval': (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P]),
    'gnx_scale_shift_relu': (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _I, _P]),
    'gnx_bn_relu_bwd': (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'gnx_bnrelu_avgpool2': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_bn_relu_bwd_pooled': (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_colsum': (_I, [_P, _L, _L, _I, _P, _I, _P, _P]),
    'gnx_masked_ce_workspace': (_L, [_L]),
    'gnx_masked_ce_fwd': (_I, [_P, _L, _P, _L, _I, _I, _F, _P, _P, _P, _P, _P]),
    'gnx_masked_ce_bwd': (_I, [_P, _L, _P, _L, _I, _I, _P, _P, _F, _P, _L, _P]),
    'gnx_meter_add': (_I, [_P, _P, _D, _P, _P, _D, _P]),
    'gnx_conv1x1_bnrelu': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _I, _I, _P]),
    'gnx_conv1x1_workspace': (_L, [_L, _I, _I]),
    'gnx_conv1x1_bnrelu_ws': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P]),
    'gnx_conv1x1_bnrelu_act': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    'gnx_conv1x1_split_pack_halves': (_L, [_I]),
    'gnx_conv1x1_split_pack': (_I, [_P, _P, _I, _P]),
    'gnx_conv1x1_bnrelu_act_split': (_I, [_P, _L, _P, _P, _L, _L, _I, _P, _P, _P, _P, _P]),
    'gnx_conv3x3_split_pack_halves': (_L, []),
    'gnx_conv3x3_split_pack': (_I, [_P, _P, _P]),
    'gnx_conv3x3_split': (_I, [_P, _L, _P, _P, _L, _L, _I, _P]),
    'gnx_wgrad1x1_split_workspace': (_L, [_L, _I, _I]),
    'gnx_wgrad1x1_split': (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _P]),
    'gnx_wgrad3x3_split_workspace': (_L, [_L]),
    'gnx_wgrad3x3_split': (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _P]),
    'gnx_conv1x1_dgrad_bn_workspace': (_L, [_L, _I]),
    'gnx_conv1x1_dgrad_bnrelu_bwd': (_I, [_P, _L, _P, _P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv1x1_dgrad_wgrad_workspace': (_L, [_L, _I]),
    'gnx_conv1x1_dgrad_wgrad_bnrelu_bwd': (_I, [_P, _L, _P, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'gnx_conv3x3_dgrad_bn_workspace': (_L, [_L, _I]),
    'gnx_conv3x3_dgrad_bnrelu_bwd': (_I, [_P, _L, _P, _P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv1x1_fold_clamp': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'gnx_conv1x1_clamped_act': (_I, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_repack_conv3x3': (_I, [_P, _P, _I, _I, _P]),
    'gnx_conv1x1_bnrelu_f16': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _I, _I, _P]),
    'gnx_conv3x3_bnrelu_f16': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P]),
    'gnx_conv1x1_bnrelu_f16_act16': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    'gnx_conv3x3_f16_dma': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P]),
    'gnx_conv_stem_bnrelu_maxpool_h16': (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'gnx_conv_stem_bnrelu_maxpool_u8': (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    'gnx_conv_stem_bnrelu_maxpool_f16mul': (_I, [_P, _I, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'gnx_u8_to_f32': (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    'gnx_conv1x1_bnrelu_f16_h': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    'gnx_conv1x1_bnrelu_h16': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    'gnx_bnrelu_avgpool2_h16': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_conv3x3_f16_dma_h': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P]),
    'gnx_bnrelu_avgpool_h16': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_wgrad1x1_f16_workspace': (_L, [_L, _I, _I]),
    'gnx_wgrad1x1_f16': (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P]),
    'gnx_wgrad3x3_f16_workspace': (_L, [_L]),
    'gnx_wgrad3x3_f16': (_I, [_P, _L, _P, _P, _P, _L, _I, _P, _I, _P, _P]),
    'gnx_conv3x3_dgrad_bnrelu_bwd_f16_workspace': (_L, [_L]),
    'gnx_conv3x3_dgrad_bnrelu_bwd_f16': (_I, [_P, _L, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv1x1_dgrad_bnrelu_bwd_f16_workspace': (_L, [_L, _I]),
    'gnx_conv1x1_dgrad_bnrelu_bwd_f16': (_I, [_P, _P, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv1x1_dgrad_wgrad_f16_workspace': (_L, [_L, _I]),
    'gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16': (_I, [_P, _P, _P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_tail_bwd_f16_workspace': (_L, [_L, _I]),
    'gnx_tail_bwd_f16': (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_trans_bwd_f16_workspace': (_L, [_L, _I, _I]),
    'gnx_trans_bwd_f16': (_I, [_P, _L, _P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_h16_cols_to_f32': (_I, [_P, _L, _P, _L, _L, _I, _P, _P, _P]),
    'gnx_stem_bwd_f16_workspace': (_L, [_L, _I]),
    'gnx_stem_bwd_f16': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _L, _I, _I, _P, _I, _P, _P]),
    'gnx_dense_layer_f16_pack': (_I, [_P, _P, _P, _P, _I, _P]),
    'gnx_dense_layer_f16': (_I, [_P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    'gnx_dense_layer_f16_tape': (_I, [_P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    'gnx_dense_layer_f16_set_form': (_I, [_I]),
    'gnx_dense_bwd_f16_pack': (_I, [_P, _P, _P, _P, _I, _P]),
    'gnx_wgrad3x3_f16_lb': (_I, [_P, _L, _P, _L, _L, _P, _P, _L, _I, _P, _I, _P, _P]),
    'gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb': (_I, [_P, _L, _P, _P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv3x3_bwd_f16_workspace': (_L, [_L]),
    'gnx_conv3x3_bwd_f16_lb': (_I, [_P, _L, _P, _P, _L, _L, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb': (_I, [_P, _P, _P, _L, _L, _P, _L, _L, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                                       _P, _P]),
    'gnx_tail_bwd_f16_lb': (_I, [_P, _L, _P, _L, _L, _P, _L, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_trans_bwd_f16_lb': (_I, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    'gnx_conv_stem_bnrelu_maxpool_f16mul_cb': (_I, [_P, _I, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'gnx_conv1x1_bnrelu_h16_cb': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P]),
    'gnx_bnrelu_avgpool2_h16_cb': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_transition_f16_pack': (_I, [_P, _P, _I, _I, _P]),
    'gnx_transition_f16': (_I, [_P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    'gnx_transition_f16_tape': (_I, [_P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _L, _P, _L, _P]),
    'gnx_bnrelu_avgpool_h16_cb': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_conv3x3_bnrelu': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P]),
    'gnx_winograd_conv3x3_weights': (_I, [_P, _P, _I, _I, _P]),
    'gnx_conv3x3_winograd': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _I, _P]),
    'gnx_conv_stem': (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_conv_stem_bnrelu_maxpool': (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'gnx_conv_stem_bnrelu_maxpool_argmax': (_I, [_P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'gnx_bnrelu_maxpool': (_I, [_P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P]),
    'gnx_bnrelu_avgpool': (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P]),
    'gnx_wgrad_workspace': (_L, [_L, _I, _I, _I]),
    'gnx_wgrad_bnrelu': (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_wgrad_bnrelu_batch': (_I, [_P, _I, _I, _P]),
    'gnx_transpose_weight': (_I, [_P, _P, _I, _I, _P]),
    'gnx_relayout_weights_batch': (_I, [_P, _I, _I, _P]),
    'gnx_repack_conv3x3_bwd': (_I, [_P, _P, _I, _I, _P]),
    'gnx_rows_broadcast': (_I, [_P, _L, _P, _L, _L, _I, _I, _F, _P]),
    'gnx_avgpool2_bwd': (_I, [_P, _L, _P, _L, _L, _I, _I, _P]),
    'gnx_maxpool_bwd': (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P]),
    'gnx_bnrelu_maxpool_argmax': (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P]),
    'gnx_maxpool_bwd_argmax': (_I, [_P, _P, _L, _P, _L, _L, _I, _I, _I, _P]),
    'gnx_maxpool_bwd_argmax_bnrelu': (_I, [_P, _P, _L, _P, _L, _P, _P, _L, _L, _I, _I, _I, _P]),
    'gnx_conv0_wgrad_workspace': (_L, [_L, _I, _I, _I, _I, _I, _I, _I]),
    'gnx_conv0_wgrad': (_I, [_P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gnx_softmax_rows': (_I, [_P, _L, _L, _I, _P, _L, _P, _P]),
    'gnx_gemm_f32': (_I, [_P, _L, _I, _P, _L, _I, _P, _P, _L, _L, _L, _L, _I, _P]),
    'gnx_gemm_f32_workspace': (_L, [_L, _L, _L]),
    'gnx_gemm_f32_ws': (_I, [_P, _L, _I, _P, _L, _I, _P, _P, _L, _L, _L, _L, _I, _P, _P]),
}

_lib = None
_ERRORS = {-1: 'bad argument', -2: 'kernel launch failed', -3: 'unsupported shape'}
ERR_UNSUPPORTED = -3


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                "gridnext_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gridnext_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=torch.float32):
    """Device pointer of a tensor that must live on a HIP device; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("gridnext_amd kernels need tensors on a HIP device (got %s); there is no CPU path"
                           % t.device)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (name, _ERRORS.get(rc, 'error'), rc))


def query(name, *args):
    return getattr(lib(), name)(*args)
