"""Host-side mirror of the reference's network API for the hot path, on top of
the C-ABI.  Names follow the reference (src/additionally.h:907-969, src/main.c:156-229):

    net = Network.from_cfg(cfg, batch, quantized)      # parse_network_cfg
    net.load_weights(path)                             # load_weights_upto_cpu
    net.fuse_conv_batchnorm()                          # yolov2_fuse_conv_batchnorm
    net.calculate_binary_weights()                     # calculate_binary_weights
    net.quantize()                                     # quantinization_and_get_multipliers
    net.to_device(0)
    out = net.predict(images)                          # network_predict_gpu_cudnn[_quantized]
    dets = net.get_boxes(image, w, h, thresh, nms=.4)  # get_network_boxes + do_nms_sort

Python is plumbing only: all arithmetic happens in libyolo2hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import YoloHipError, check, lib

LAYER_TYPES = {0: "conv", 3: "maxpool", 8: "route", 13: "shortcut", 21: "region", 22: "yolo",
               23: "upsample", 24: "reorg", 25: "blank"}
INFO_FIELDS = ("type", "batch", "w", "h", "c", "n", "size", "stride", "pad", "out_w", "out_h", "out_c",
               "outputs", "inputs", "activation", "xnor", "int8", "index", "classes", "coords", "total",
               "softmax", "conv_mode", "batch_normalize")


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_float_p)


class Network:
    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self._on_device = False

    # ------------------------------------------------------------ build
    @classmethod
    def from_cfg(cls, cfg_path: str, batch: int = 1, quantized: int = 0) -> "Network":
        h = C.c_void_p()
        check(lib.yl_network_create_from_cfg(cfg_path.encode(), batch, quantized, C.byref(h)),
              "yl_network_create_from_cfg")
        return cls(h)

    @classmethod
    def from_desc(cls, descs, batch: int, w: int, h: int, c: int, quantized: int = 0,
                  input_calibration: Optional[np.ndarray] = None) -> "Network":
        arr = (_lib.LayerDesc * len(descs))(*descs)
        hd = C.c_void_p()
        ic = None
        n_ic = 0
        if input_calibration is not None and len(input_calibration):
            ic_arr = np.ascontiguousarray(input_calibration, dtype=np.float32)
            ic, n_ic = _fp(ic_arr), len(ic_arr)
        check(lib.yl_network_create_from_desc(arr, len(descs), batch, w, h, c, quantized, ic, n_ic, C.byref(hd)),
              "yl_network_create_from_desc")
        return cls(hd)

    @classmethod
    def load(cls, cfg_path: str, weights_path: str, batch: int = 1, quantized: int = 0,
             device: Optional[int] = None, debug: bool = False, fuse: bool = False,
             quant_rule: int = 0, winograd: bool = True, bf16: bool = False, device_prep: bool = False,
             variant: Optional[int] = None, device_pack: Optional[bool] = None, strict: bool = False,
             split_k: bool = False) -> "Network":
        """The full prep sequence of test_detector_cpu (src/main.c:160-171)."""
        net = cls.from_cfg(cfg_path, batch, quantized)
        if quant_rule:
            net.set_quant_rule(quant_rule)
        if not winograd:
            check(lib.yl_network_set_winograd(net._h, 0), "yl_network_set_winograd")
        if bf16:
            net.set_precision(1)
        if strict:
            net.set_precision(2)
        if split_k:
            check(lib.yl_network_set_split_k(net._h, 1), "yl_network_set_split_k")
        if variant is not None:
            net.set_variant(variant)       # before to_device: bit 5 selects the Winograd weight packing
        if device_pack is not None:
            check(lib.yl_network_set_device_pack(net._h, 1 if device_pack else 0), "yl_network_set_device_pack")
        net.load_weights(weights_path)
        if device_prep:
            # the same three passes on the GPU, bit-identical results (csrc/prep.hip)
            check(lib.yl_network_prepare_on_device(net._h, device if device is not None else 0),
                  "yl_network_prepare_on_device")
        else:
            net.fuse_conv_batchnorm()
            net.calculate_binary_weights()
            if quantized:
                net.quantize()
        if debug:
            check(lib.yl_network_set_debug(net._h, 1), "yl_network_set_debug")
        if fuse:
            net.set_fusion(True)
        if device is not None:
            net.to_device(device)
        return net

    def load_weights(self, path: str) -> None:
        check(lib.yl_network_load_weights(self._h, path.encode()), "yl_network_load_weights")

    def fuse_conv_batchnorm(self) -> None:
        check(lib.yl_network_fuse_conv_batchnorm(self._h), "yl_network_fuse_conv_batchnorm")

    def calculate_binary_weights(self) -> None:
        check(lib.yl_network_calculate_binary_weights(self._h), "yl_network_calculate_binary_weights")

    def quantize(self) -> None:
        check(lib.yl_network_quantize(self._h), "yl_network_quantize")

    def close(self) -> None:
        if self._h:
            lib.yl_network_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ introspection
    @property
    def n(self) -> int:
        return lib.yl_network_num_layers(self._h)

    @property
    def batch(self) -> int:
        return lib.yl_network_batch(self._h)

    @property
    def input_dims(self):
        d = (C.c_int * 3)()
        check(lib.yl_network_input_dims(self._h, d), "yl_network_input_dims")
        return d[0], d[1], d[2]          # w, h, c

    def layer_info(self, i: int) -> dict:
        info = (C.c_int * 24)()
        check(lib.yl_network_layer_info(self._h, i, info), "yl_network_layer_info")
        return dict(zip(INFO_FIELDS, list(info)))

    def layers(self) -> List[dict]:
        return [self.layer_info(i) for i in range(self.n)]

    def _arr(self, ptr, n, dtype):
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    def layer_weights(self, i: int):
        li = self.layer_info(i)
        return self._arr(lib.yl_network_layer_weights(self._h, i), li["n"] * li["c"] * li["size"] ** 2, np.float32)

    def layer_biases(self, i: int):
        return self._arr(lib.yl_network_layer_biases(self._h, i), self.layer_info(i)["n"], np.float32)

    def layer_weights_int8(self, i: int):
        li = self.layer_info(i)
        return self._arr(lib.yl_network_layer_weights_int8(self._h, i), li["n"] * li["c"] * li["size"] ** 2, np.int8)

    def layer_mean_arr(self, i: int):
        return self._arr(lib.yl_network_layer_mean_arr(self._h, i), self.layer_info(i)["n"], np.float32)

    def layer_quant_multipliers(self, i: int):
        m = (C.c_float * 2)()
        check(lib.yl_network_layer_quant_multipliers(self._h, i, m), "yl_network_layer_quant_multipliers")
        return float(m[0]), float(m[1])

    def layer_traffic(self, i: int):
        """(read, written) algorithmic HBM bytes of layer i per forward under the fusion plan"""
        b = (C.c_double * 2)()
        check(lib.yl_network_layer_traffic(self._h, i, b), "yl_network_layer_traffic")
        return float(b[0]), float(b[1])

    @property
    def flops_per_image(self) -> float:
        return lib.yl_network_flops_per_image(self._h)

    def layer_head(self, i: int):
        """(mask, anchors) of YOLO/REGION layer i"""
        na = C.c_int(0)
        n = lib.yl_network_layer_head(self._h, i, None, 0, None, 0, C.byref(na))        # sizes first
        if n < 0:
            raise YoloHipError("yl_network_layer_head failed: " + _lib.last_error())
        mask = (C.c_int * max(n, 1))()
        anchors = (C.c_float * max(na.value, 1))()
        if lib.yl_network_layer_head(self._h, i, mask, n, anchors, na.value, None) < 0:
            raise YoloHipError("yl_network_layer_head failed: " + _lib.last_error())
        return np.array(mask[:n], dtype=np.int32), np.array(anchors[:na.value], dtype=np.float32)

    def layer_tree(self, i: int):
        """(parent[classes], group_size[groups]) of a REGION layer with a softmax tree, or None"""
        groups = lib.yl_network_layer_tree(self._h, i, None, None)
        if groups < 0:
            raise YoloHipError("yl_network_layer_tree failed: " + _lib.last_error())
        if groups == 0:
            return None
        parent = (C.c_int * self.layer_info(i)["classes"])()
        gs = (C.c_int * groups)()
        lib.yl_network_layer_tree(self._h, i, parent, gs)
        return np.array(parent[:], dtype=np.int32), np.array(gs[:], dtype=np.int32)

    # ------------------------------------------------------------ device
    def set_quant_rule(self, rule: int) -> None:
        """0 = the reference CPU path's INT8 layer set, 1 = its GPU path's (`l.quantized`); before to_device"""
        check(lib.yl_network_set_quant_rule(self._h, rule), "yl_network_set_quant_rule")

    def set_conv_tile(self, cfg: int) -> None:
        check(lib.yl_network_set_conv_tile(self._h, cfg), "yl_network_set_conv_tile")

    def set_precision(self, precision: int) -> None:
        """0 = FP32 (default), 1 = opt-in BF16 operands for the FP32 convolutions, 2 = strict FP32 (FP32-MFMA direct kernels
        only); before to_device"""
        check(lib.yl_network_set_precision(self._h, precision), "yl_network_set_precision")

    def set_variant(self, bits: int) -> None:
        check(lib.yl_network_set_variant(self._h, bits), "yl_network_set_variant")

    def set_int8_tile(self, cfg: int) -> None:
        check(lib.yl_network_set_int8_tile(self._h, cfg), "yl_network_set_int8_tile")

    def set_split_k(self, on: bool = True) -> None:
        """K ranges for FP32 convolutions whose grid leaves CUs idle (deterministic two-stage sum); before to_device"""
        check(lib.yl_network_set_split_k(self._h, 1 if on else 0), "yl_network_set_split_k")

    def set_nms_mode(self, mode: int) -> None:
        check(lib.yl_network_set_nms_mode(self._h, mode), "yl_network_set_nms_mode")

    def set_fusion(self, on: bool = True) -> None:
        """Fold same-shape linear [shortcut] layers into the preceding conv's epilogue (before to_device)."""
        check(lib.yl_network_set_fusion(self._h, 1 if on else 0), "yl_network_set_fusion")

    def to_device(self, device: int = 0) -> None:
        check(lib.yl_network_to_device(self._h, device), "yl_network_to_device")
        self._on_device = True

    def set_stream(self, stream_ptr: int) -> None:
        check(lib.yl_network_set_stream(self._h, C.c_void_p(stream_ptr)), "yl_network_set_stream")

    def synchronize(self) -> None:
        check(lib.yl_network_synchronize(self._h), "yl_network_synchronize")

    def predict(self, images: np.ndarray) -> np.ndarray:
        """images: float32 [batch, c, h, w] in [0,1]; returns the last layer's output (a copy)."""
        w, h, c = self.input_dims
        x = np.ascontiguousarray(images, dtype=np.float32)
        if x.size != self.batch * c * h * w:
            raise ValueError("input has %d elements, network wants %d" % (x.size, self.batch * c * h * w))
        p = lib.yl_network_predict(self._h, _fp(x))
        if not p:
            raise YoloHipError("yl_network_predict failed: " + _lib.last_error())
        last = self.layer_info(self.n - 1)
        return np.ctypeslib.as_array(p, shape=(self.batch * last["outputs"],)).copy()

    def predict_raw(self, images: np.ndarray) -> None:
        """yl_network_predict on a host batch without copying the returned tensor out (timing the boundary itself)"""
        x = np.ascontiguousarray(images, dtype=np.float32)
        if not lib.yl_network_predict(self._h, _fp(x)):
            raise YoloHipError("yl_network_predict failed: " + _lib.last_error())

    def forward_device(self, input_dev_ptr: int) -> None:
        check(lib.yl_network_forward(self._h, C.c_void_p(input_dev_ptr)), "yl_network_forward")

    @property
    def input_dev(self) -> int:
        return lib.yl_network_input_dev(self._h)

    def layer_output_dev(self, i: int) -> int:
        return lib.yl_network_layer_output_dev(self._h, i)

    def layer_output(self, i: int) -> np.ndarray:
        li = self.layer_info(i)
        out = np.empty(self.batch * li["outputs"], dtype=np.float32)
        check(lib.yl_network_layer_output(self._h, i, _fp(out)), "yl_network_layer_output")
        return out

    def layer_output_image(self, i: int, image: int) -> np.ndarray:
        li = self.layer_info(i)
        out = np.empty(li["outputs"], dtype=np.float32)
        check(lib.yl_network_layer_output_image(self._h, i, image, _fp(out)), "yl_network_layer_output_image")
        return out

    def layer_packed(self, i: int, which: int) -> Optional[np.ndarray]:
        """the packed weight image of conv layer i on the device as bytes (which: 0 k-major FP32, 1 Winograd U,
        2 int8 / bf16 units, 3 XNOR sign words, 4 XNOR count thresholds int32[Mpad + 1], 5 / 6 XNOR mean / bias
        float32[M], 7 the three-piece bf16 weights of conv_f32_x3.hip); None if the layer has none"""
        n = lib.yl_debug_layer_packed(self._h, i, which, None, 0)
        if n < 0:
            raise YoloHipError("yl_debug_layer_packed failed: " + _lib.last_error())
        if n == 0:
            return None
        out = np.empty(n, dtype=np.uint8)
        if lib.yl_debug_layer_packed(self._h, i, which, out.ctypes.data_as(C.c_void_p), n) != n:
            raise YoloHipError("yl_debug_layer_packed failed: " + _lib.last_error())
        return out

    def layer_materialised(self, i: int) -> bool:
        """False for FP32 tensors the fusion plan never writes (folded conv, int8-only consumer)"""
        return bool(lib.yl_network_layer_output_dev(self._h, i))

    def layer_xnor_counts(self, i: int) -> np.ndarray:
        li = self.layer_info(i)
        out = np.empty(self.batch * li["outputs"], dtype=np.int32)
        check(lib.yl_network_layer_xnor_counts(self._h, i, out.ctypes.data_as(_lib.c_int32_p)),
              "yl_network_layer_xnor_counts")
        return out

    def layer_int8_acc(self, i: int) -> np.ndarray:
        li = self.layer_info(i)
        out = np.empty(self.batch * li["outputs"], dtype=np.int32)
        check(lib.yl_network_layer_int8_acc(self._h, i, out.ctypes.data_as(_lib.c_int32_p)),
              "yl_network_layer_int8_acc")
        return out

    def profile(self, input_dev_ptr: int, iters: int = 5):
        ms = np.zeros(self.n, dtype=np.float32)
        tot = C.c_float(0)
        check(lib.yl_network_profile(self._h, C.c_void_p(input_dev_ptr), iters, _fp(ms), C.byref(tot)),
              "yl_network_profile")
        return ms, float(tot.value)

    def forward_timed(self, input_dev_ptr: int, slot: int = 0) -> None:
        check(lib.yl_network_forward_timed(self._h, C.c_void_p(input_dev_ptr), slot), "yl_network_forward_timed")

    def layer_times(self, slot: int = 0):
        ms = np.zeros(self.n, dtype=np.float32)
        tot = C.c_float(0)
        check(lib.yl_network_layer_times(self._h, slot, _fp(ms), C.byref(tot)), "yl_network_layer_times")
        return ms, float(tot.value)

    def layer_kernel(self, i: int) -> str:
        p = lib.yl_network_layer_kernel(self._h, i)
        return p.decode() if p else ""

    def set_input_u8(self, image: int, pixels: np.ndarray) -> None:
        """pixels: uint8 [h][w][c] as a decoder delivers them; converts (/255.), resizes
        (resize_image) and stores into batch slot `image` of the device input buffer."""
        pix = np.ascontiguousarray(pixels, dtype=np.uint8)
        if pix.ndim != 3:
            raise ValueError("pixels must be [h][w][c]")
        h, w, c = pix.shape
        check(lib.yl_network_set_input_u8(self._h, image, pix.ctypes.data_as(C.c_void_p), w, h, c),
              "yl_network_set_input_u8")

    def set_input_u8_batch(self, frames, first: int = 0) -> None:
        """frames: a list of uint8 [h][w][c] arrays -> batch slots first, first + 1, ... in ONE call (pool-copied, uploaded on the copy
        stream, resized on the compute stream); same bits as one set_input_u8 per frame"""
        pix = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        n = len(pix)
        if n == 0:
            return
        if any(p.ndim != 3 for p in pix):
            raise ValueError("every frame must be [h][w][c]")
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pix])
        ws = (C.c_int * n)(*[p.shape[1] for p in pix])
        hs = (C.c_int * n)(*[p.shape[0] for p in pix])
        check(lib.yl_network_set_input_u8_batch(self._h, first, n, ptrs, ws, hs, pix[0].shape[2]), "yl_network_set_input_u8_batch")

    def set_input_u8_dev(self, image: int, pixels_dev_ptr: int, w: int, h: int, c: int = 3) -> None:
        check(lib.yl_network_set_input_u8_dev(self._h, image, C.c_void_p(pixels_dev_ptr), w, h, c),
              "yl_network_set_input_u8_dev")

    def input_download(self) -> np.ndarray:
        w, h, c = self.input_dims
        out = np.zeros((self.batch, c, h, w), dtype=np.float32)
        check(lib.yl_network_input_download(self._h, _fp(out)), "yl_network_input_download")
        return out

    def forward_staged(self) -> None:
        """forward over the device input buffer the set_input_* calls filled (asynchronous)"""
        check(lib.yl_network_forward(self._h, lib.yl_network_input_dev(self._h)), "yl_network_forward")

    def calibrate(self, images: np.ndarray) -> np.ndarray:
        """`darknet detector calibrate`: input multipliers (one per conv layer) from float32 images
        [n, c, h, w] in [0,1], n a multiple of the batch; the network must be FP32 and on the device."""
        x = np.ascontiguousarray(images, dtype=np.float32)
        w, h, c = self.input_dims
        n_img = x.size // (c * h * w)
        out = np.zeros(self.n, dtype=np.float32)
        k = lib.yl_network_calibrate(self._h, _fp(x), n_img, _fp(out), self.n)
        if k < 0:
            raise YoloHipError("yl_network_calibrate failed: " + _lib.last_error())
        return out[:k]

    # ------------------------------------------------------------ detections
    def pull_heads(self) -> None:
        check(lib.yl_network_pull_heads(self._h), "yl_network_pull_heads")

    def get_boxes(self, image: int, w: int, h: int, thresh: float, nms: float = 0.0,
                  relative: int = 1, letter: int = 0, max_rows: int = 4096) -> np.ndarray:
        classes = self.layer_info(self.n - 1)["classes"]
        max_rows = min(max_rows, 4096)          # YL_DETECT_MAX_CAP
        # count first (the batch's decode is cached in the library), then exactly the rows that exist
        n = lib.yl_network_get_boxes(self._h, image, w, h, thresh, relative, letter, nms, None, 0, None)
        if n < 0:
            raise YoloHipError("yl_network_get_boxes failed: " + _lib.last_error())
        k = min(n, max_rows)
        rows = np.zeros((k, 6 + classes), dtype=np.float32)
        if k and lib.yl_network_get_boxes(self._h, image, w, h, thresh, relative, letter, nms, _fp(rows), k, None) < 0:
            raise YoloHipError("yl_network_get_boxes failed: " + _lib.last_error())
        return rows

    @staticmethod
    def _dims(sizes, batch):
        if sizes is None:
            return None, None
        wh = np.ascontiguousarray(np.broadcast_to(np.asarray(sizes, dtype=np.int32).reshape(-1, 2), (batch, 2)))
        w = np.ascontiguousarray(wh[:, 0])
        h = np.ascontiguousarray(wh[:, 1])
        return w, h

    def detect_batch(self, thresh: float, nms: float, cap: int, records_dev_ptr: int, counts_dev_ptr: int,
                     sizes=None, relative: int = 1, letter: int = 0) -> None:
        """get_network_boxes + do_nms_sort for every image, on the GPU, into device buffers
        (asynchronous).  sizes = (w, h) or [(w, h)] * batch of the source images, None = relative
        to the network input."""
        w, h = self._dims(sizes, self.batch)
        ip = C.POINTER(C.c_int)
        check(lib.yl_network_detect_batch(self._h, w.ctypes.data_as(ip) if w is not None else None,
                                          h.ctypes.data_as(ip) if h is not None else None, thresh, relative, letter,
                                          nms, cap, C.c_void_p(records_dev_ptr), C.c_void_p(counts_dev_ptr)),
              "yl_network_detect_batch")

    def get_boxes_batch(self, thresh: float, nms: float = 0.0, cap: int = 1024, sizes=None, relative: int = 1,
                        letter: int = 0):
        """list (one entry per image) of row arrays [n][6+classes] -- the batched, on-GPU form of
        get_boxes(); `counts` (second result) may exceed cap, in which case rows are truncated."""
        w, h = self._dims(sizes, self.batch)
        ip = C.POINTER(C.c_int)
        classes = self.layer_info(self.n - 1)["classes"]
        rows = np.zeros((self.batch, cap, 6 + classes), dtype=np.float32)
        counts = np.zeros(self.batch, dtype=np.int32)
        check(lib.yl_network_get_boxes_batch(self._h, w.ctypes.data_as(ip) if w is not None else None,
                                             h.ctypes.data_as(ip) if h is not None else None, thresh, relative,
                                             letter, nms, cap, _fp(rows), counts.ctypes.data_as(ip)),
              "yl_network_get_boxes_batch")
        return [rows[b, :min(int(counts[b]), cap)] for b in range(self.batch)], counts

    def compact_detections(self, thresh: float, cap: int, records_dev_ptr: int, counts_dev_ptr: int) -> None:
        check(lib.yl_network_compact_detections(self._h, thresh, cap, C.c_void_p(records_dev_ptr),
                                                C.c_void_p(counts_dev_ptr)), "yl_network_compact_detections")
