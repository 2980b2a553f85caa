"""ctypes binding of libmiseg.so (C ABI declared in include/miseg.h).

The library is the only compute path of this package: if it is missing or the
device is not a gfx950 the import of the engine fails loudly; there is no CPU
or eager fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISEG_LIB") or os.path.join(_HERE, "libmiseg.so")   # MISEG_LIB: experiments

OK = 0
E_INVALID_ARG, E_UNSUPPORTED_SHAPE, E_HIP, E_NO_DEVICE = -1, -2, -3, -4
DT_BF16, DT_F32 = 0, 1
FMT_BF16, FMT_F16 = 0, 1      # 16-bit storage format of activations and weights (miseg_conv_desc.fmt)
ACT_NONE, ACT_RELU, ACT_SOFTMAX, ACT_SIGMOID = 0, 1, 2, 3
LOSS_CCE, LOSS_DICE, LOSS_FOCAL = 1, 2, 4
RES_ADD, RES_BN_RELU_MASK = 0, 1
TILE_NO_PERSIST = 128          # miseg_conv_desc.tile_hint flag: keep a layer off its persistent kernel


class MisegError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """Mirror of `miseg_conv_desc`."""
    _fields_ = [
        ("inp", C.c_void_p), ("weight", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
        ("residual", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("scale2", C.c_void_p), ("shift2", C.c_void_p), ("in_scale", C.c_void_p),
        ("in_shift", C.c_void_p), ("collapse_out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride_y", C.c_int32), ("stride_x", C.c_int32),
        ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("in_pix_stride", C.c_int32), ("in_row_stride", C.c_int32), ("in_img_stride", C.c_int64),
        ("out_pix_stride", C.c_int32), ("out_row_stride", C.c_int32), ("out_img_stride", C.c_int64),
        ("out2_pix_stride", C.c_int32), ("out2_row_stride", C.c_int32), ("out2_img_stride", C.c_int64),
        ("res_pix_stride", C.c_int32), ("res_row_stride", C.c_int32), ("res_img_stride", C.c_int64),
        ("out_dtype", C.c_int32), ("res_dtype", C.c_int32), ("act", C.c_int32),
        ("transposed", C.c_int32), ("temperature", C.c_float), ("collapse_value", C.c_int32),
        ("tile_hint", C.c_int32),
        ("stats", C.c_void_p), ("stats_replicas", C.c_int32), ("stats_stride", C.c_int32),
        ("fmt", C.c_int32), ("res_after_act", C.c_int32), ("res_upsample_shift", C.c_int32),
        ("out2_linear", C.c_int32),
        ("in_valid_c", C.c_int32),
        ("res_mode", C.c_int32), ("res_mask_c", C.c_int32),
    ]


class LinknetTailDesc(C.Structure):
    """Mirror of `miseg_linknet_tail_desc`."""
    _fields_ = [
        ("inp", C.c_void_p), ("wa", C.c_void_p), ("wb", C.c_void_p), ("wc", C.c_void_p), ("wd", C.c_void_p),
        ("sa", C.c_void_p), ("ta", C.c_void_p), ("sb", C.c_void_p), ("tb", C.c_void_p),
        ("sc", C.c_void_p), ("tc", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("collapse_out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("in_pix_stride", C.c_int32), ("in_row_stride", C.c_int32), ("in_img_stride", C.c_int64),
        ("out_pix_stride", C.c_int32), ("out_row_stride", C.c_int32), ("out_img_stride", C.c_int64),
        ("classes", C.c_int32), ("act", C.c_int32), ("temperature", C.c_float),
        ("collapse_value", C.c_int32), ("fmt", C.c_int32),
    ]


class LinknetBlockDesc(C.Structure):
    """Mirror of `miseg_linknet_block_desc`."""
    _fields_ = [
        ("inp", C.c_void_p), ("wa", C.c_void_p), ("wb", C.c_void_p), ("wc", C.c_void_p),
        ("sa", C.c_void_p), ("ta", C.c_void_p), ("sb", C.c_void_p), ("tb", C.c_void_p),
        ("sc", C.c_void_p), ("tc", C.c_void_p), ("skip", C.c_void_p), ("out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("in_pix_stride", C.c_int32), ("in_row_stride", C.c_int32), ("in_img_stride", C.c_int64),
        ("skip_pix_stride", C.c_int32), ("skip_row_stride", C.c_int32), ("skip_img_stride", C.c_int64),
        ("out_pix_stride", C.c_int32), ("out_row_stride", C.c_int32), ("out_img_stride", C.c_int64),
        ("fmt", C.c_int32), ("cin", C.c_int32),
    ]


class FpnHeadDesc(C.Structure):
    """Mirror of `miseg_fpn_head_desc`."""
    _fields_ = [
        ("inp", C.c_void_p), ("wd", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("collapse_out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("in_pix_stride", C.c_int32), ("in_row_stride", C.c_int32), ("in_img_stride", C.c_int64),
        ("out_pix_stride", C.c_int32), ("out_row_stride", C.c_int32), ("out_img_stride", C.c_int64),
        ("classes", C.c_int32), ("act", C.c_int32), ("temperature", C.c_float),
        ("collapse_value", C.c_int32), ("fmt", C.c_int32),
    ]


class WgradDesc(C.Structure):
    """Mirror of `miseg_wgrad_desc`."""
    _fields_ = [
        ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride_y", C.c_int32), ("stride_x", C.c_int32),
        ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("x_pix", C.c_int32), ("x_row", C.c_int32), ("x_img", C.c_int64),
        ("dy_pix", C.c_int32), ("dy_row", C.c_int32), ("dy_img", C.c_int64),
        ("tile_hint", C.c_int32),
        ("x_scale", C.c_void_p), ("x_shift", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/miseg.h declares
_VP, _I32, _I64, _FP = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
SIGNATURES = {
    "miseg_last_error": (C.c_char_p, []),
    "miseg_version": (C.c_int, []),
    "miseg_device_check": (C.c_int, []),
    "miseg_conv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), _VP]),
    "miseg_conv_kernel_name": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, _I32]),
    "miseg_conv_c32_head_fwd": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _VP]),
    "miseg_dense_layer_fwd": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _VP]),
    "miseg_linknet_tail_fwd": (C.c_int, [C.POINTER(LinknetTailDesc), _VP]),
    "miseg_linknet_block_fwd": (C.c_int, [C.POINTER(LinknetBlockDesc), _VP]),
    "miseg_fpn_head_fwd": (C.c_int, [C.POINTER(FpnHeadDesc), _VP]),
    "miseg_prep_input": (C.c_int, [_FP, _VP, _FP, _FP, _I32, _I32, _I32, _VP]),
    "miseg_prep_input_indicator": (C.c_int, [_FP, _VP, _FP, _FP, _I32, _I32, _I32, _VP]),
    "miseg_maxpool3x3s2_fwd": (C.c_int, [_VP, _VP, _FP, _FP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_avgpool2x2_fwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_nearest_fwd": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_bilinear2x_fwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_scale_shift_relu_fwd": (C.c_int, [_VP, _VP, _FP, _FP, _I64, _I32, _I32, _I32, _VP]),
    "miseg_prep_input_f16": (C.c_int, [_FP, _VP, _FP, _FP, _I32, _I32, _I32, _VP]),
    "miseg_maxpool3x3s2_fwd_f16": (C.c_int, [_VP, _VP, _FP, _FP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_avgpool2x2_fwd_f16": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_nearest_fwd_f16": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_bilinear2x_fwd_f16": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_scale_shift_relu_fwd_f16": (C.c_int, [_VP, _VP, _FP, _FP, _I64, _I32, _I32, _I32, _VP]),
    "miseg_collapse_u8": (C.c_int, [_FP, _VP, _I64, _I32, _I32, _VP]),
    "miseg_loss_sums": (C.c_int, [_FP, _FP, _VP, _I64, _I32, _VP]),
    "miseg_conv2d_wgrad": (C.c_int, [C.POINTER(WgradDesc), _VP]),
    "miseg_bn_stats_fwd": (C.c_int, [_VP, _I64, _I32, _I32, _FP, _FP, _FP, _FP, C.c_float, C.c_float,
                                     _FP, _FP, _FP, _FP, _VP, _VP]),
    "miseg_bn_stats_fwd_noalloc": (C.c_int, [_VP, _I64, _I32, _I32, _FP, _FP, _FP, _FP, C.c_float, C.c_float,
                                             _FP, _FP, _FP, _FP, _VP]),
    "miseg_bn_finalize": (C.c_int, [_VP, _I32, _I32, _I64, _I32, _FP, _FP, _FP, _FP, C.c_float, C.c_float,
                                    _FP, _FP, _FP, _FP, _VP]),
    "miseg_input_stats_fwd": (C.c_int, [_FP, _I64, _FP, _FP, _FP, C.c_float, C.c_float, _FP, _FP, _FP,
                                        _FP, _VP, _VP]),
    "miseg_bn_relu_bwd_reduce": (C.c_int, [_VP, _VP, _I64, _I32, _I32, _I32, _FP, _FP, _FP, _FP,
                                           C.c_float, _I32, _VP, _VP]),
    "miseg_bn_bwd_sums": (C.c_int, [_VP, _I32, _I32, _I32, _FP, _FP, C.c_float, _VP, _VP]),
    "miseg_bn_relu_bwd_apply": (C.c_int, [_VP, _VP, _VP, _I64, _I32, _I32, _I32, _I32, _FP, _FP, _FP,
                                          _FP, C.c_float, _I32, _VP, _VP, _I32, _VP]),
    "miseg_maxpool3x3s2_bwd": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _I32, _VP]),
    "miseg_avgpool2x2_bwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_maxpool2x2_fwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_maxpool2x2_bwd": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_im2col3x3_fwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_relu_bwd": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_nearest_bwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_upsample_bilinear2x_bwd": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_copy_channels": (C.c_int, [_VP, _VP, _VP, _I64, _I32, _I32, _I32, _I32, _VP]),
    "miseg_channel_sums": (C.c_int, [_VP, _I64, _I32, _I32, _VP, _VP]),
    "miseg_loss_grad": (C.c_int, [_FP, _FP, _VP, _VP, _I64, _I32, _I32, C.c_float, C.c_float, _I32,
                                  _I32, _VP]),
    "miseg_loss_grad_ex": (C.c_int, [_FP, _FP, _VP, _VP, _I64, _I32, _I32, C.c_float, C.c_float, _I32,
                                     C.c_float, C.c_float, _VP]),
    "miseg_loss_sums_focal": (C.c_int, [_FP, _FP, _VP, _I64, _I32, C.c_float, C.c_float, _VP]),
    "miseg_adam_step": (C.c_int, [_FP, _FP, _FP, _FP, _VP, _I64, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_float, _VP]),
    "miseg_image_minmax": (C.c_int, [_VP, _I32, _I32, _I64, _FP, _VP]),
    "miseg_image_normalize": (C.c_int, [_VP, _I32, _I32, _I64, _FP, _FP, _VP]),
    "miseg_resize_bilinear": (C.c_int, [_VP, _I32, _FP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_resize_nearest_u8": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_one_hot": (C.c_int, [_VP, _FP, _I64, _I32, _VP]),
    "miseg_augment": (C.c_int, [_FP, _FP, _FP, _FP, _I32, _I32, _I32, _I32, _FP, _VP, _VP]),
    "miseg_comm_unique_id": (C.c_int, [_VP]),
    "miseg_comm_init": (C.c_int, [C.POINTER(_VP), _I32, _I32, _VP]),
    "miseg_allreduce_async": (C.c_int, [_VP, _FP, _I64, _VP]),
    "miseg_comm_destroy": (C.c_int, [_VP]),
    "miseg_plan_create": (_VP, []),
    "miseg_plan_destroy": (None, [_VP]),
    "miseg_plan_num_ops": (C.c_int, [_VP]),
    "miseg_plan_set_format": (C.c_int, [_VP, _I32]),
    "miseg_plan_add_conv": (C.c_int, [_VP, C.POINTER(ConvDesc)]),
    "miseg_plan_add_conv_c32_head": (C.c_int, [_VP, C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "miseg_plan_add_dense_layer": (C.c_int, [_VP, C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    "miseg_plan_add_linknet_tail": (C.c_int, [_VP, C.POINTER(LinknetTailDesc)]),
    "miseg_plan_add_linknet_block": (C.c_int, [_VP, C.POINTER(LinknetBlockDesc)]),
    "miseg_plan_add_fpn_head": (C.c_int, [_VP, C.POINTER(FpnHeadDesc)]),
    "miseg_plan_add_prep_input": (C.c_int, [_VP, _FP, _VP, _FP, _FP, _I32, _I32, _I32]),
    "miseg_plan_add_maxpool3x3s2": (C.c_int, [_VP, _VP, _VP, _FP, _FP, _I32, _I32, _I32, _I32, _I32, _I32]),
    "miseg_plan_add_avgpool2x2": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32]),
    "miseg_plan_add_maxpool2x2": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32]),
    "miseg_plan_add_im2col3x3": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32]),
    "miseg_plan_add_head_gather": (C.c_int, [_VP, _VP, _FP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, C.c_float,
                                             _I32]),
    "miseg_head_gather_fwd": (C.c_int, [_VP, _FP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, C.c_float, _I32, _I32,
                                        _VP]),
    "miseg_head_gather_bwd": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I32, _I32, _I32, _VP]),
    "miseg_plan_add_upsample_nearest": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "miseg_plan_add_upsample_bilinear2x": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32]),
    "miseg_plan_add_scale_shift_relu": (C.c_int, [_VP, _VP, _VP, _FP, _FP, _I64, _I32, _I32, _I32]),
    "miseg_plan_run": (C.c_int, [_VP, _VP]),
    "miseg_plan_run_range": (C.c_int, [_VP, C.c_int, C.c_int, _VP, C.POINTER(C.c_float)]),
    "miseg_plan_op_macs": (C.c_int64, [_VP, C.c_int]),
    "miseg_plan_op_kernel": (C.c_int, [_VP, C.c_int, C.c_char_p, _I32]),
}

_lib = None


def load():
    """Load libmiseg.so and bind every declared symbol.  Raises MisegError when
    the library has not been built (run `python __graft_entry__.py build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MisegError(
            "libmiseg.so not found at %s: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path."
            % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().miseg_last_error()
        raise MisegError("libmiseg error %d: %s" % (rc, (msg or b"").decode()))
