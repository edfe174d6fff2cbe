"""ctypes binding of libaddk.so (the C ABI declared in include/addk.h).

The product path has no CPU fallback: if the shared library is missing or a
symbol is absent, importing/using `addk` raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADDK_LIB') or os.path.join(_HERE, 'libaddk.so')      # ADDK_LIB: an A/B or diagnostic build of the same sources (scripts/*.sh)

MAX_SRC, MAX_SLAB, MAX_TERMS = 12, 32, 4
fp = C.POINTER(C.c_float)
i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class Src(C.Structure):
    _fields_ = [('x', vp), ('a', vp), ('b', vp), ('ld', i32), ('C', i32), ('relu', i32), ('rs_hw', i32)]


class ConvArgs(C.Structure):
    _fields_ = [('src', Src * MAX_SRC), ('nsrc', i32), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('Cout', i32), ('ldw', i32),
                ('cin_total', i32), ('w_choff', i32), ('ldy', i32), ('w', vp), ('y', vp), ('bias', vp), ('bias_n', vp),
                ('stats', vp), ('stats_ld', i32), ('_pad', i32), ('wpack', vp), ('wpack_floats', i64), ('wpack_ready', i32), ('_pad2', i32),
                ('rs_y', vp), ('rs_ldy', i32), ('_pad3', i32)]


class ConvDgradArgs(C.Structure):
    _fields_ = [('dy', vp), ('lddy', i32), ('Cout', i32), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('w', vp), ('ldw', i32),
                ('cin_total', i32), ('w_choff', i32), ('dst', Src), ('g', vp), ('ldg', i32), ('accumulate', i32),
                ('dab', vp), ('wpack', vp), ('wpack_floats', i64), ('wpack_ready', i32), ('_pad2', i32)]


class ConvWgradArgs(C.Structure):
    _fields_ = [('dy', vp), ('lddy', i32), ('Cout', i32), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('src', Src), ('dw', vp),
                ('ldw', i32), ('cin_total', i32), ('w_choff', i32), ('accumulate', i32), ('ws', vp), ('ws_floats', i64)]


class BnFinalizeArgs(C.Structure):
    _fields_ = [('partial', vp), ('rows', i32), ('C', i32), ('count', f64), ('gamma', vp), ('beta', vp),
                ('running_mean', vp), ('running_var', vp), ('momentum', f32), ('eps', f32), ('a', vp), ('b', vp),
                ('mean', vp), ('invstd', vp)]


class SepArgs(C.Structure):
    _fields_ = [('src', Src), ('N', i32), ('H', i32), ('W', i32), ('K', i32), ('Cout', i32), ('ldw', i32), ('dw_w', vp), ('pw_w', vp),
                ('y', vp), ('ldy', i32), ('ldt', i32), ('t', vp), ('stats', vp), ('stats_ld', i32), ('stats_rows', i32), ('nterm', i32), ('ea', vp), ('eb', vp),
                ('term', Src * MAX_TERMS), ('fin', BnFinalizeArgs), ('fin_counter', vp)]


class SepBwdArgs(C.Structure):
    _fields_ = [('dy', vp), ('lddy', i32), ('N', i32), ('H', i32), ('W', i32), ('K', i32), ('src', Src), ('Cout', i32), ('ldw', i32),
                ('dw_w', vp), ('pw_w', vp), ('g', vp), ('ldg', i32), ('accumulate', i32), ('dab', vp), ('ws', vp)]


class CeUpsampleArgs(C.Structure):
    _fields_ = [('logits', vp), ('ld', i32), ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('OH', i32), ('OW', i32),
                ('target', vp), ('class_w', vp), ('ignore_index', i32), ('wsum', vp), ('scale', f32), ('loss_out', vp),
                ('g', vp), ('ldg', i32), ('accumulate', i32), ('ws', vp)]


class DwArgs(C.Structure):
    _fields_ = [('src', Src), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32), ('KH', i32), ('KW', i32),
                ('stride', i32), ('pad', i32), ('dil', i32), ('w', vp), ('y', vp), ('ldy', i32)]


class DwBwdArgs(C.Structure):
    _fields_ = [('dy', vp), ('lddy', i32), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32), ('KH', i32),
                ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('src', Src), ('w', vp), ('g', vp), ('ldg', i32),
                ('accumulate', i32), ('dab', vp), ('dw', vp), ('dw_accumulate', i32), ('ws', vp), ('defer_wreduce', i32), ('_pad', i32)]


class DwWreduceItem(C.Structure):
    _fields_ = [('ws', vp), ('dw', vp), ('rows', i32), ('n', i32), ('accumulate', i32), ('_pad', i32)]


class BnEvalEntry(C.Structure):
    _fields_ = [('gamma', vp), ('beta', vp), ('rm', vp), ('rv', vp), ('a', vp), ('b', vp), ('C', i32), ('eps', f32)]


class BnBwdArgs(C.Structure):
    _fields_ = [('slab', vp * MAX_SLAB), ('rows', i32 * MAX_SLAB), ('nslab', i32), ('C', i32), ('count', f64),
                ('gamma', vp), ('mean', vp), ('invstd', vp), ('a', vp), ('dgamma', vp), ('dbeta', vp),
                ('accumulate', i32), ('c1', vp), ('c2', vp), ('dmv', vp), ('centered', i32), ('_pad', i32)]


class SlabReduceItem(C.Structure):
    _fields_ = [('partial', vp), ('out', vp), ('rows', i32), ('C', i32)]


class BnCoeffsItem(C.Structure):
    _fields_ = [('dmv', vp), ('c1', vp), ('c2', vp), ('count', f64), ('C', i32), ('_pad', i32)]


class BnApplyItem(C.Structure):
    _fields_ = [('g', vp), ('x', vp), ('c1', vp), ('c2', vp), ('mean', vp), ('out', vp), ('P', i64), ('ldg', i32), ('ldx', i32), ('ldo', i32),
                ('C', i32)]


class AffineSumArgs(C.Structure):
    _fields_ = [('term', Src * MAX_TERMS), ('nterm', i32), ('P', i64), ('C', i32), ('out', vp), ('ldo', i32),
                ('relu_out', i32), ('accumulate', i32)]


class AffineSumBwdArgs(C.Structure):
    _fields_ = [('term', Src * MAX_TERMS), ('nterm', i32), ('P', i64), ('C', i32), ('dout', vp), ('lddo', i32),
                ('out', vp), ('ldo', i32), ('relu_out', i32), ('g', vp * MAX_TERMS), ('ldg', i32 * MAX_TERMS),
                ('accumulate', i32 * MAX_TERMS), ('dab', vp * MAX_TERMS)]


class EdmArgs(C.Structure):
    _fields_ = [('src', Src), ('N', i32), ('H', i32), ('W', i32), ('ldo', i32), ('conv_w', vp), ('w1', vp), ('b1', vp), ('w2', vp), ('b2', vp),
                ('w3', vp), ('b3', vp), ('out', vp), ('out_host', vp), ('ws', vp)]


class ResizeArgs(C.Structure):
    _fields_ = [('src', Src), ('N', i32), ('H', i32), ('W', i32), ('OH', i32), ('OW', i32), ('y', vp), ('ldy', i32),
                ('nchw_out', i32)]


class ResizeBwdArgs(C.Structure):
    _fields_ = [('dy', vp), ('lddy', i32), ('nchw_in', i32), ('dy_scale', vp), ('src', Src), ('N', i32), ('H', i32),
                ('W', i32), ('OH', i32), ('OW', i32), ('g', vp), ('ldg', i32), ('accumulate', i32), ('dab', vp)]


_SIGS = {
    'addk_last_error': (C.c_char_p, []),
    'addk_version': (i32, []),
    'addk_selftest_mfma': (i32, [vp, vp]),
    'addk_conv_fwd': (i32, [C.POINTER(ConvArgs), vp]),
    'addk_conv_rows': (i32, [i64, i32]),
    'addk_set_conv_precision': (i32, [i32]),
    'addk_set_split_min_channels': (i32, [i32]),
    'addk_get_conv_precision': (i32, []),
    'addk_conv_dgrad': (i32, [C.POINTER(ConvDgradArgs), vp]),
    'addk_set_fast_paths': (i32, [i32]),
    'addk_get_fast_paths': (i32, []),
    'addk_conv_fwd_pack_floats': (i64, [C.POINTER(ConvArgs)]),
    'addk_conv_dgrad_pack_floats': (i64, [C.POINTER(ConvDgradArgs)]),
    'addk_conv_fwd_batch_key': (i32, [C.POINTER(ConvArgs)]),
    'addk_conv_dgrad_batch_key': (i32, [C.POINTER(ConvDgradArgs)]),
    'addk_conv_fwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_conv_dgrad_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_conv_batch_run': (i32, [vp, vp, vp]),
    'addk_conv_pack_desc_bytes': (i64, []),
    'addk_conv_fwd_pack_desc': (i32, [C.POINTER(ConvArgs), vp]),
    'addk_conv_dgrad_pack_desc': (i32, [C.POINTER(ConvDgradArgs), vp]),
    'addk_conv_pack_batch': (i32, [vp, i32, vp]),
    'addk_conv_fwd_resample_ok': (i32, [C.POINTER(ConvArgs)]),
    'addk_debug_trace_fatal_signals': (i32, []),
    'addk_comm_mailbox_bytes': (i64, [i32, i64]),
    'addk_comm_alloc': (i32, [i32, i64, C.POINTER(vp), vp]),
    'addk_comm_open': (i32, [i32, i32, i64, vp, vp, C.POINTER(vp)]),
    'addk_comm_allreduce': (i32, [vp, vp, i64, i32, vp]),
    'addk_comm_status': (i32, [vp, C.POINTER(i64), C.POINTER(i64)]),
    'addk_comm_close': (i32, [vp, vp]),
    'addk_edm_head_ws_bytes': (i64, [i32, i32, i32]),
    'addk_edm_head_supported': (i32, [C.POINTER(EdmArgs)]),
    'addk_edm_head': (i32, [C.POINTER(EdmArgs), vp]),
    'addk_conv_wgrad': (i32, [C.POINTER(ConvWgradArgs), vp]),
    'addk_conv_wgrad_ws': (i64, [i64, i32, i32, i32]),
    'addk_conv_wgrad_config': (i32, [C.POINTER(ConvWgradArgs), C.POINTER(i32)]),
    'addk_conv_wgrad_batch_prepare': (i64, [C.POINTER(ConvWgradArgs), i32, vp, i64, C.POINTER(i64)]),
    'addk_conv_wgrad_batch_run': (i32, [vp, C.POINTER(i64), vp]),
    'addk_lut_u8': (i32, [vp, vp, i64, vp, vp]),
    'addk_resample_u8': (i32, [vp, i32, i32, vp, i32, i32, i32, vp, vp, i32, i32, i32, vp]),
    'addk_nearest_u8': (i32, [vp, i32, i32, vp, i32, i32, vp, vp, i32, vp]),
    'addk_finish_sample': (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'addk_sep_fwd_supported': (i32, [C.POINTER(SepArgs)]),
    'addk_sep_rows': (i32, [C.POINTER(SepArgs)]),
    'addk_bn_fin_ws_bytes': (i64, [i32, i32]),
    'addk_sep_bwd_rows': (i32, [C.POINTER(SepBwdArgs)]),
    'addk_sep_bwd': (i32, [C.POINTER(SepBwdArgs), vp]),
    'addk_sep_bwd_batch_key': (i32, [C.POINTER(SepBwdArgs)]),
    'addk_sep_bwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_sep_bwd_batch_run': (i32, [vp, vp, vp]),
    'addk_sep_fwd': (i32, [C.POINTER(SepArgs), vp]),
    'addk_sep_fwd_batch_key': (i32, [C.POINTER(SepArgs)]),
    'addk_sep_fwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_sep_batch_run': (i32, [vp, vp, vp]),
    'addk_dw_fwd': (i32, [C.POINTER(DwArgs), vp]),
    'addk_dw_bwd': (i32, [C.POINTER(DwBwdArgs), vp]),
    'addk_dw_rows': (i32, [i64, i32]),
    'addk_dw_wreduce_batch': (i32, [vp, i32, vp]),
    'addk_dw_fwd_batch_key': (i32, [C.POINTER(DwArgs)]),
    'addk_dw_bwd_batch_key': (i32, [C.POINTER(DwBwdArgs)]),
    'addk_dw_fwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_dw_bwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_dw_batch_run': (i32, [vp, vp, vp]),
    'addk_bn_finalize_batch': (i32, [vp, i32, i32, vp]),
    'addk_slab_reduce_batch': (i32, [vp, i32, i32, vp]),
    'addk_bn_bwd_coeffs_batch': (i32, [vp, i32, i32, vp]),
    'addk_bn_bwd_batch': (i32, [vp, i32, i32, vp]),
    'addk_bn_bwd_apply_batch': (i32, [vp, i32, i64, vp]),
    'addk_bn_finalize': (i32, [C.POINTER(BnFinalizeArgs), vp]),
    'addk_slab_reduce': (i32, [vp, i32, i32, vp, vp]),
    'addk_bn_eval_affine': (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp]),
    'addk_bn_eval_affine_batch': (i32, [vp, i32, vp]),
    'addk_bn_bwd': (i32, [C.POINTER(BnBwdArgs), vp]),
    'addk_bn_bwd_coeffs_from_dmv': (i32, [vp, i32, f64, vp, vp, vp]),
    'addk_affine_sum_fwd': (i32, [C.POINTER(AffineSumArgs), vp]),
    'addk_affine_sum_bwd': (i32, [C.POINTER(AffineSumBwdArgs), vp]),
    'addk_ew_rows': (i32, [i64, i32]),
    'addk_bn_bwd_apply': (i32, [vp, i32, vp, i32, vp, vp, vp, i64, i32, vp, i32, vp]),
    'addk_resize_fwd': (i32, [C.POINTER(ResizeArgs), vp]),
    'addk_resize_bwd': (i32, [C.POINTER(ResizeBwdArgs), vp]),
    'addk_resize_bwd_batch_key': (i32, [C.POINTER(ResizeBwdArgs)]),
    'addk_resize_bwd_batch_prepare': (i64, [vp, i32, vp, i64, vp]),
    'addk_resize_bwd_batch_run': (i32, [vp, vp, vp]),
    'addk_gap_fwd': (i32, [C.POINTER(Src), i32, i32, vp, i32, vp, i32, vp]),
    'addk_gap_bwd': (i32, [C.POINTER(Src), i32, i32, vp, i32, vp, i32, i32, vp, vp]),
    'addk_pool3_fwd': (i32, [C.POINTER(Src), i32, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    'addk_pool3_bwd': (i32, [C.POINTER(Src), i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, vp]),
    'addk_nchw_to_nhwc': (i32, [vp, i32, i32, i64, vp, i32, vp]),
    'addk_nhwc_to_nchw': (i32, [C.POINTER(Src), i32, i64, vp, vp]),
    'addk_nchw_grad_to_nhwc': (i32, [vp, C.POINTER(Src), i32, i64, vp, i32, i32, vp, vp]),
    'addk_ce_count': (i32, [vp, i64, vp, i32, i32, vp, vp, vp]),
    'addk_ce_fwd_bwd': (i32, [vp, vp, i32, i32, i64, vp, i32, vp, f32, vp, vp, vp, vp]),
    'addk_ce_ws_floats': (i64, [i32, i64]),
    'addk_ce_upsample_supported': (i32, [i32, i32, i32, i32, i32, i32]),
    'addk_ce_upsample_ws_floats': (i64, [i32, i32, i32]),
    'addk_ce_upsample_fwd_bwd': (i32, [C.POINTER(CeUpsampleArgs), vp]),
    'addk_sgd_step': (i32, [vp, vp, vp, i64, vp, f32, f32, i32, i32, f32, vp]),
    'addk_fill': (i32, [vp, i64, f32, vp]),
    'addk_entropy_sum': (i32, [vp, i32, i32, i64, vp, vp, vp]),
    'addk_argmax_nchw': (i32, [vp, i32, i32, i64, vp, vp]),
    'addk_confusion': (i32, [vp, vp, i64, i32, vp, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


class AddkError(RuntimeError):
    pass


def load():
    """Load libaddk.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AddkError('libaddk.so not found at %s — run `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950).  addk has no CPU fallback.' % LIB_PATH)
    # libaddk.so needs libamdhip64.so.7; PyTorch bundles its own copy under the same soname.  Import torch FIRST so that the
    # process has one HIP runtime (torch's): loaded the other way round, launches on torch's streams fail with
    # hipErrorNoDevice ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        raise AddkError('%s failed (%d): %s' % (what or 'addk call', rc, load().addk_last_error().decode()))
