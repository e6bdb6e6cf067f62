"""ctypes binding of libvltk_hip.so (include/vltk_hip.h).

This is the reference-side stub a vltk maintainer would add:
plain pointers and sizes over the C ABI, no torch types.  There is NO CPU
fallback: if the library is missing or a GPU is not present the import / call
fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvltk_hip.so")
# tools only: the ablation build (make ABLATION=1 OUT=.../libvltk_hip_ablation.so) -- stamp / timing-only kernels, never the product
if os.environ.get("VLTK_AMD_LIB"):              # tools only: A/B of two builds on one box (e.g. the previous commit's library)
    LIB_PATH = os.environ["VLTK_AMD_LIB"]
elif os.environ.get("VLTK_AMD_ABLATION_LIB") == "1":
    LIB_PATH = LIB_PATH.replace("libvltk_hip.so", "libvltk_hip_ablation.so")

VK_OK, VK_EINVAL, VK_ENOTIMPL, VK_ENONFINITE, VK_EWEIGHTS, VK_EHIP, VK_ENOMEM = range(7)
VK_F32, VK_F16, VK_I64, VK_I32, VK_BF16 = 0, 1, 2, 3, 4
VK_ACT_NONE, VK_ACT_RELU, VK_ACT_GELU, VK_ACT_TANH = 0, 1, 2, 3
VK_MAX_ANCHOR_DIM = 8
VK_MAX_NMS_THRESH = 8

# status code -> the Python exception type the reference raises in the same situation
# (frcnn.py:1930 NotImplementedError, :148 AssertionError, :1789/:1850 EnvironmentError/OSError)
_EXC = {
    VK_EINVAL: ValueError,
    VK_ENOTIMPL: NotImplementedError,
    VK_ENONFINITE: AssertionError,
    VK_EWEIGHTS: OSError,
    VK_EHIP: RuntimeError,
    VK_ENOMEM: MemoryError,
}


class vk_config(C.Structure):
    _fields_ = [
        ("depth", C.c_int32), ("num_groups", C.c_int32), ("width_per_group", C.c_int32),
        ("stem_out_channels", C.c_int32), ("res2_out_channels", C.c_int32),
        ("stride_in_1x1", C.c_int32), ("caffe_maxpool", C.c_int32),
        ("num_sizes", C.c_int32), ("sizes", C.c_float * VK_MAX_ANCHOR_DIM),
        ("num_ratios", C.c_int32), ("ratios", C.c_float * VK_MAX_ANCHOR_DIM),
        ("anchor_offset", C.c_float), ("rpn_hidden_channels", C.c_int32),
        ("rpn_min_size", C.c_float), ("rpn_nms_thresh", C.c_double),
        ("pre_nms_topk", C.c_int32), ("post_nms_topk", C.c_int32),
        ("rpn_bbox_weights", C.c_float * 4),
        ("num_classes", C.c_int32), ("num_attrs", C.c_int32), ("use_attr", C.c_int32),
        ("pooler_resolution", C.c_int32), ("res5_halve", C.c_int32),
        ("cls_agnostic_bbox_reg", C.c_int32), ("roi_bbox_weights", C.c_float * 4),
        ("precision", C.c_int32),
    ]


class vk_roi_params(C.Structure):
    _fields_ = [
        ("num_nms_thresh", C.c_int32), ("nms_thresh", C.c_double * VK_MAX_NMS_THRESH),
        ("min_detections", C.c_int32), ("max_detections", C.c_int32),
    ]


class vk_outputs(C.Structure):
    _fields_ = [
        ("obj_ids", C.c_void_p), ("obj_probs", C.c_void_p), ("attr_ids", C.c_void_p),
        ("attr_probs", C.c_void_p), ("boxes", C.c_void_p), ("preds_per_image", C.c_void_p),
        ("roi_features", C.c_void_p),
    ]


_P, _I, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes): every symbol include/vltk_hip.h declares
SIGNATURES = {
    "vk_last_error": (C.c_char_p, []),
    "vk_version": (_I, []),
    "vk_create": (_I, [C.POINTER(vk_config), _I, C.POINTER(_P)]),
    "vk_load_weights": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _I]),
    "vk_finalize": (_I, [_P]),
    "vk_destroy": (_I, [_P]),
    "vk_set_option": (_I, [_P, C.c_char_p, _I]),
    "vk_num_weights": (_I, [_P, C.POINTER(_I)]),
    "vk_weight_name": (_I, [_P, _I, C.POINTER(C.c_char_p)]),
    "vk_forward": (_I, [_P, _P, _I, _I, _I, _P, _P, C.POINTER(vk_roi_params), C.POINTER(vk_outputs), _P]),
    "vk_forward_begin": (_I, [_P, _P, _I, _I, _I, _P, _P, C.POINTER(vk_roi_params), C.POINTER(vk_outputs), _P,
                              C.POINTER(C.c_int64)]),
    "vk_forward_end": (_I, [_P, C.c_int64]),
    "vk_get_stage": (_I, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_I), C.POINTER(C.c_int64), C.POINTER(_I)]),
    "vk_memcpy_d2d": (_I, [_P, _P, _SZ, _P]),
    "vk_enable_stage_timing": (_I, [_P, _I]),
    "vk_get_stage_timing": (_I, [_P, C.POINTER(_F)]),
    "vk_enable_kernel_timing": (_I, [_P, _I]),
    "vk_get_kernel_timing": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), _I]),
    "vk_packed_weight_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "vk_conv_slice_channels": (_I, [_I, _I]),
    "vk_packed_cout": (_I, [_I]),
    "vk_pack_conv_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vk_conv1x1_dual": (_I, [_P, _I, _P, _I, C.c_long, _P, _P, _P, _P, _I, _I, _P]),
    "vk_bottleneck64": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vk_conv1x1_meanpool_workspace_bytes": (_SZ, [_I, _I, _I]),
    "vk_conv1x1_meanpool": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _SZ, _P]),
    "vk_linear": (_I, [_P, C.c_long, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vk_layernorm": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _P]),
    "vk_embed_layernorm": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _F, _I, _P]),
    "vk_attention": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vk_roi_align": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "vk_assign_levels": (_I, [_P, _I, _I, _I, _I, _F, _I, _P, _P]),
    "vk_upsample2x_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vk_subsample2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vk_relu_copy": (_I, [_P, _P, C.c_long, _I, _P]),
    "vk_rpn_multilevel_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "vk_rpn_proposals_multilevel": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _F, _P, _P, _F, _D, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "vk_conv2d": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vk_nchw_to_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "vk_nhwc_to_nchw": (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "vk_packed_stem_bytes": (_SZ, [_I, _I]),
    "vk_pack_stem_weight": (_I, [_P, _P, _I, _I, _P, _P]),
    "vk_stem": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _SZ, _P]),
    "vk_stem_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "vk_stem_out_hw": (None, [_I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "vk_preprocess": (_I, [_P, _P, _P, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _F, _P, _P]),
    "vk_maxpool3x3s2": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "vk_rpn_workspace_bytes": (_SZ, [_I, _I, _I]),
    "vk_rpn_proposals": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _F, _P, C.POINTER(_F), _F, _D, _I, _I,
                              _P, _P, _P, _P, _P, _SZ, _P]),
    "vk_nms": (_I, [_P, _P, _I, _D, _P, _P, _P, _SZ, _P]),
    "vk_nms_workspace_bytes": (_SZ, [_I]),
    "vk_roi_pool": (_I, [_P, _I, _I, _I, _I, _P, _I, _F, _I, _P, _I, _P]),
    "vk_mean_pool": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "vk_box_decode": (_I, [_P, _P, _I, _I, C.POINTER(_F), _P, _P]),
    "vk_make_rois": (_I, [_P, _I, _I, _P, _P]),
    "vk_softmax_argmax": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "vk_concat_embed": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P]),
    "vk_chosen_deltas": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "vk_roi_outputs": (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, C.POINTER(_F),
                            C.POINTER(vk_roi_params), C.POINTER(vk_outputs), _P, _P, _P]),
}

_lib = None


def load():
    """Load the shared library and set every prototype.  Raises if it is missing -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vltk_amd/csrc`).  vltk_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status):
    if status != VK_OK:
        msg = load().vk_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(msg)


def call(name, *args):
    """Call an int-status entry point and map failures to the reference's exception types."""
    check(getattr(load(), name)(*args))
