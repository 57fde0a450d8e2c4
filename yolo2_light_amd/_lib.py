"""ctypes binding of libyolo2hip.so (the C-ABI declared in include/yolo2_hip.h).

There is deliberately no fallback: if the shared library is missing the import
fails loudly, and every device call raises when HIP reports an error.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO2HIP_LIB: another build of the SAME library (A/B runs of tools/); never a fallback
LIB_PATH = os.environ.get("YOLO2HIP_LIB") or os.path.join(_HERE, "libyolo2hip.so")


class YoloHipError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C yolo2_light_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_int8_p = C.POINTER(C.c_int8)
c_int32_p = C.POINTER(C.c_int32)


class LayerDesc(C.Structure):
    """struct yl_layer_desc (include/yolo2_hip.h)."""
    _fields_ = [
        ("type", C.c_int), ("activation", C.c_int),
        ("batch", C.c_int), ("w", C.c_int), ("h", C.c_int), ("c", C.c_int),
        ("n", C.c_int), ("size", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("out_w", C.c_int), ("out_h", C.c_int), ("out_c", C.c_int),
        ("outputs", C.c_int), ("inputs", C.c_int),
        ("batch_normalize", C.c_int), ("xnor", C.c_int), ("quantized", C.c_int), ("index", C.c_int),
        ("input_layers", c_int_p), ("input_sizes", c_int_p),
        ("classes", C.c_int), ("coords", C.c_int), ("total", C.c_int), ("softmax", C.c_int),
        ("mask", c_int_p), ("anchors", c_float_p), ("scale", C.c_float),
        ("weights", c_float_p), ("biases", c_float_p),
        ("scales", c_float_p), ("rolling_mean", c_float_p), ("rolling_variance", c_float_p),
        ("weights_int8", c_int8_p),
        ("input_quant_multipler", C.c_float), ("weights_quant_multipler", C.c_float),
        ("mean_arr", c_float_p),
        ("output", c_float_p),
        ("tree_n", C.c_int), ("tree_groups", C.c_int), ("tree_parent", c_int_p), ("tree_group_size", c_int_p),
    ]


_vp = C.c_void_p

ABI_VERSION = 4          # YL_ABI_VERSION of the include/yolo2_hip.h this table was written against

_SIGS = {
    "yl_abi_version": (C.c_int, []),
    "yl_last_error": (C.c_char_p, []),
    "yl_device_count": (C.c_int, []),
    "yl_device_synchronize": (C.c_int, [C.c_int]),
    "yl_network_create_from_cfg": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]),
    "yl_network_create_from_desc": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, c_float_p, C.c_int, C.POINTER(_vp)]),
    "yl_network_load_weights": (C.c_int, [_vp, C.c_char_p]),
    "yl_network_fuse_conv_batchnorm": (C.c_int, [_vp]),
    "yl_network_calculate_binary_weights": (C.c_int, [_vp]),
    "yl_network_quantize": (C.c_int, [_vp]),
    "yl_network_destroy": (None, [_vp]),
    "yl_network_num_layers": (C.c_int, [_vp]),
    "yl_network_batch": (C.c_int, [_vp]),
    "yl_network_input_dims": (C.c_int, [_vp, c_int_p]),
    "yl_network_layer_info": (C.c_int, [_vp, C.c_int, c_int_p]),
    "yl_network_layer_weights": (c_float_p, [_vp, C.c_int]),
    "yl_network_layer_biases": (c_float_p, [_vp, C.c_int]),
    "yl_network_layer_weights_int8": (c_int8_p, [_vp, C.c_int]),
    "yl_network_layer_mean_arr": (c_float_p, [_vp, C.c_int]),
    "yl_network_layer_quant_multipliers": (C.c_int, [_vp, C.c_int, c_float_p]),
    "yl_network_flops_per_image": (C.c_double, [_vp]),
    "yl_network_layer_traffic": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    "yl_network_to_device": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_fusion": (C.c_int, [_vp, C.c_int]),
    "yl_network_predict": (c_float_p, [_vp, c_float_p]),
    "yl_network_forward": (C.c_int, [_vp, _vp]),
    "yl_network_set_stream": (C.c_int, [_vp, _vp]),
    "yl_network_synchronize": (C.c_int, [_vp]),
    "yl_network_layer_output": (C.c_int, [_vp, C.c_int, c_float_p]),
    "yl_network_layer_output_image": (C.c_int, [_vp, C.c_int, C.c_int, c_float_p]),
    "yl_network_layer_output_dev": (_vp, [_vp, C.c_int]),
    "yl_network_input_dev": (_vp, [_vp]),
    "yl_network_set_debug": (C.c_int, [_vp, C.c_int]),
    "yl_network_layer_xnor_counts": (C.c_int, [_vp, C.c_int, c_int32_p]),
    "yl_network_layer_int8_acc": (C.c_int, [_vp, C.c_int, c_int32_p]),
    "yl_network_profile": (C.c_int, [_vp, _vp, C.c_int, c_float_p, c_float_p]),
    "yl_network_forward_timed": (C.c_int, [_vp, _vp, C.c_int]),
    "yl_network_layer_times": (C.c_int, [_vp, C.c_int, c_float_p, c_float_p]),
    "yl_network_layer_kernel": (C.c_char_p, [_vp, C.c_int]),
    "yl_network_get_boxes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float,
                                       c_float_p, C.c_int, c_int_p]),
    "yl_network_pull_heads": (C.c_int, [_vp]),
    "yl_network_set_conv_tile": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_int8_tile": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_winograd": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_variant": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_precision": (C.c_int, [_vp, C.c_int]),
    "yl_network_load_weights_upto": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "yl_network_prepare_on_device": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_nms_mode": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_split_k": (C.c_int, [_vp, C.c_int]),
    "yl_network_set_device_pack": (C.c_int, [_vp, C.c_int]),
    "yl_debug_layer_packed": (C.c_longlong, [_vp, C.c_int, C.c_int, _vp, C.c_longlong]),
    "yl_network_set_quant_rule": (C.c_int, [_vp, C.c_int]),
    "yl_network_layer_head": (C.c_int, [_vp, C.c_int, c_int_p, C.c_int, c_float_p, C.c_int, c_int_p]),
    "yl_network_layer_tree": (C.c_int, [_vp, C.c_int, c_int_p, c_int_p]),
    "yl_debug_wino_pack": (C.c_longlong, [c_float_p, C.c_int, C.c_int, C.c_int, c_float_p, C.c_longlong]),
    "yl_debug_x3_pack": (C.c_longlong, [c_float_p, C.c_int, C.c_int, C.c_int, _vp, C.c_longlong]),
    "yl_debug_row3_pack": (C.c_longlong, [c_float_p, C.c_int, C.c_int, _vp, C.c_longlong]),
    "yl_network_compact_detections": (C.c_int, [_vp, C.c_float, C.c_int, _vp, _vp]),
    "yl_network_detect_batch": (C.c_int, [_vp, c_int_p, c_int_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                                          _vp, _vp]),
    "yl_network_set_input_u8": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "yl_network_set_input_u8_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "yl_network_set_input_u8_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "yl_network_input_download": (C.c_int, [_vp, c_float_p]),
    "yl_network_calibrate": (C.c_int, [_vp, c_float_p, C.c_int, c_float_p, C.c_int]),
    "yl_entropy_from_histogram": (C.c_float, [C.POINTER(C.c_uint32), C.c_int, C.c_float]),
    "yl_network_get_boxes_batch": (C.c_int, [_vp, c_int_p, c_int_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                                             c_float_p, c_int_p]),
    "yl_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p]),
    "yl_group_create": (C.c_int, [_vp, c_int_p, C.c_int, C.POINTER(_vp)]),
    "yl_group_destroy": (None, [_vp]),
    "yl_group_size": (C.c_int, [_vp]),
    "yl_group_shard": (C.c_int, [_vp, C.c_int, c_int_p, c_int_p]),
    "yl_group_member": (_vp, [_vp, C.c_int]),
    "yl_group_predict": (c_float_p, [_vp, c_float_p]),
    "yl_group_forward": (C.c_int, [_vp, C.POINTER(_vp)]),
    "yl_group_synchronize": (C.c_int, [_vp]),
    "yl_group_detect_batch": (C.c_int, [_vp, c_int_p, c_int_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, _vp, _vp]),
    "yl_group_get_boxes_batch": (C.c_int, [_vp, c_int_p, c_int_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                                           c_float_p, c_int_p]),
}

# every symbol include/yolo2_hip.h declares (tests/test_abi.py cross-checks against the header)
EXPORTED = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here == missing export == hard failure
    _fn.restype = _res
    _fn.argtypes = _args


if lib.yl_abi_version() != ABI_VERSION:
    raise ImportError("%s implements C-ABI version %d, this binding was written against %d (include/yolo2_hip.h)"
                      % (LIB_PATH, lib.yl_abi_version(), ABI_VERSION))


def last_error() -> str:
    return (lib.yl_last_error() or b"").decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise YoloHipError("%s failed (%d): %s" % (what, rc, last_error()))
