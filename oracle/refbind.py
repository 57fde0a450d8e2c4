"""ctypes binding of oracle/_ref/libyolo2ref*.so -- TEST INFRASTRUCTURE.

The libraries are the UNMODIFIED reference CPU path compiled by oracle/Makefile
from /root/reference/src plus oracle/ref_shim.c.  Only tests/, the
__graft_entry__.smoke() check and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(_HERE, "_ref", "libyolo2ref.so")
FAST = os.path.join(_HERE, "_ref", "libyolo2ref_fast.so")
# reference host code + integration/network_predict_hip.c bound to libyolo2hip.so (drop-in test)
HIP = os.path.join(_HERE, "_ref", "libyolo2ref_hip.so")

INFO_FIELDS = ("type", "batch", "w", "h", "c", "n", "size", "stride", "pad", "out_w", "out_h", "out_c",
               "outputs", "inputs", "activation", "xnor", "quantized", "index", "classes", "coords", "total",
               "softmax", "new_lda", "batch_normalize")

_fp = C.POINTER(C.c_float)


def available(fast: bool = False) -> bool:
    return os.path.exists(FAST if fast else GOLD)


def _bind(path: str) -> C.CDLL:
    lib = C.CDLL(path)      # RTLD_LOCAL: the two builds export the same symbols
    vp = C.c_void_p
    lib.ref_load.restype = vp
    lib.ref_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    lib.ref_predict.restype = _fp
    lib.ref_predict.argtypes = [vp, _fp]
    lib.ref_time_predict.restype = C.c_double
    lib.ref_time_predict.argtypes = [vp, _fp, C.c_int]
    for name in ("ref_num_layers", "ref_batch", "ref_net_w", "ref_net_h", "ref_net_c"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [vp]
    lib.ref_layer_info.restype = None
    lib.ref_layer_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    for name in ("ref_layer_output", "ref_layer_weights", "ref_layer_biases", "ref_layer_mean_arr",
                 "ref_layer_binary_weights", "ref_layer_anchors"):
        getattr(lib, name).restype = _fp
        getattr(lib, name).argtypes = [vp, C.c_int]
    lib.ref_layer_weights_int8.restype = C.POINTER(C.c_int8)
    lib.ref_layer_weights_int8.argtypes = [vp, C.c_int]
    lib.ref_layer_input_mult.restype = C.c_float
    lib.ref_layer_input_mult.argtypes = [vp, C.c_int]
    lib.ref_layer_weights_mult.restype = C.c_float
    lib.ref_layer_weights_mult.argtypes = [vp, C.c_int]
    lib.ref_layer_route_inputs.restype = C.POINTER(C.c_int)
    lib.ref_layer_route_inputs.argtypes = [vp, C.c_int]
    lib.ref_layer_mask.restype = C.POINTER(C.c_int)
    lib.ref_layer_mask.argtypes = [vp, C.c_int]
    lib.ref_get_detections.restype = C.c_int
    lib.ref_get_detections.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _fp, C.c_int,
                                       C.POINTER(C.c_int)]
    lib.ref_maxpool.restype = None
    lib.ref_maxpool.argtypes = [_fp, _fp] + [C.c_int] * 9
    lib.ref_entropy_calibration.restype = C.c_float
    lib.ref_entropy_calibration.argtypes = [_fp, C.c_size_t, C.c_float, C.c_int]
    lib.ref_load_resized.restype = C.c_int
    lib.ref_load_resized.argtypes = [C.c_char_p, C.c_int, C.c_int, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ref_set_quiet.restype = None
    lib.ref_set_quiet.argtypes = [C.c_int]
    return lib


class RefNetwork:
    """The reference's own network, driven through the shim (src/main.c:160-229 sequence)."""

    def __init__(self, cfg: str, weights: str, batch: int = 1, quantized: int = 0, fast: bool = False,
                 hip: bool = False):
        self.lib = _bind(HIP if hip else (FAST if fast else GOLD))
        if hip:
            self.lib.ref_predict_hip.restype = _fp
            self.lib.ref_predict_hip.argtypes = [C.c_void_p, _fp]
            self.lib.ref_free_hip.restype = None
        self.h = self.lib.ref_load(cfg.encode(), (weights or "").encode(), batch, quantized)
        if not self.h:
            raise RuntimeError("ref_load failed")
        self.quantized = quantized
        self.n = self.lib.ref_num_layers(self.h)
        self.batch = self.lib.ref_batch(self.h)
        self.w, self.hgt, self.c = (self.lib.ref_net_w(self.h), self.lib.ref_net_h(self.h),
                                    self.lib.ref_net_c(self.h))

    def layer_info(self, i: int) -> dict:
        info = (C.c_int * 24)()
        self.lib.ref_layer_info(self.h, i, info)
        return dict(zip(INFO_FIELDS, list(info)))

    def predict(self, x: np.ndarray) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        self._keep = x
        self.lib.ref_predict(self.h, x.ctypes.data_as(_fp))

    def predict_hip(self, x: np.ndarray) -> None:
        """network_predict_hip(net, input): the reference's host code driving libyolo2hip.so."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        self._keep = x
        if not self.lib.ref_predict_hip(self.h, x.ctypes.data_as(_fp)):
            raise RuntimeError("network_predict_hip returned NULL")

    def time_predict(self, x: np.ndarray, iters: int) -> float:
        x = np.ascontiguousarray(x, dtype=np.float32)
        return self.lib.ref_time_predict(self.h, x.ctypes.data_as(_fp), iters)

    def layer_output(self, i: int) -> np.ndarray:
        li = self.layer_info(i)
        p = self.lib.ref_layer_output(self.h, i)
        return np.ctypeslib.as_array(p, shape=(self.batch * li["outputs"],)).copy()

    def _conv_arr(self, fn, i, dtype=np.float32, per_filter=False):
        li = self.layer_info(i)
        n = li["n"] if per_filter else li["n"] * li["c"] * li["size"] ** 2
        p = fn(self.h, i)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)

    def layer_weights(self, i): return self._conv_arr(self.lib.ref_layer_weights, i)
    def layer_biases(self, i): return self._conv_arr(self.lib.ref_layer_biases, i, per_filter=True)
    def layer_weights_int8(self, i): return self._conv_arr(self.lib.ref_layer_weights_int8, i, np.int8)
    def layer_mean_arr(self, i): return self._conv_arr(self.lib.ref_layer_mean_arr, i, per_filter=True)

    def layer_quant_multipliers(self, i):
        return float(self.lib.ref_layer_input_mult(self.h, i)), float(self.lib.ref_layer_weights_mult(self.h, i))

    def get_detections(self, image: int, w: int, h: int, thresh: float, nms: float = 0.0, relative: int = 1,
                       max_rows: int = 200000) -> np.ndarray:
        classes = C.c_int(0)
        last = self.layer_info(self.n - 1)
        rows = np.zeros((max_rows, 6 + last["classes"]), dtype=np.float32)
        n = self.lib.ref_get_detections(self.h, image, w, h, thresh, nms, relative,
                                        rows.ctypes.data_as(_fp), max_rows, C.byref(classes))
        return rows[:min(n, max_rows)].copy()
