"""Shared test helpers: oracle bindings, a python network walker that runs the
oracle restatement layer by layer, seeded inputs, comparison helpers.

Everything here is test infrastructure; the product never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import refbind  # noqa: E402
from yolo2_light_amd import Network, weights, zoo  # noqa: E402

ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_fp = C.POINTER(C.c_float)
_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)

# layer type / activation codes (== reference enums)
CONV, MAXPOOL, ROUTE, SHORTCUT, REGION, YOLO, UPSAMPLE, REORG = 0, 3, 8, 13, 21, 22, 23, 24
CONV_F32, CONV_INT8, CONV_XNOR = 0, 1, 2


class OracleHead(C.Structure):
    """struct oracle_head (oracle/detect_oracle.c)"""
    _fields_ = [("type", C.c_int), ("w", C.c_int), ("h", C.c_int), ("n", C.c_int), ("classes", C.c_int),
                ("outputs", C.c_int), ("output", _fp), ("mask", C.POINTER(C.c_int)), ("anchors", _fp),
                ("tree_parent", C.POINTER(C.c_int))]


def _build_oracle() -> None:
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("yolo2_oracle.c", "detect_oracle.c")]
    if os.path.exists(ORACLE_SO) and all(os.path.getmtime(ORACLE_SO) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"],
                          stdout=subprocess.DEVNULL)


def oracle_lib() -> C.CDLL:
    _build_oracle()
    lib = C.CDLL(ORACLE_SO)
    i = C.c_int
    lib.oracle_conv_f32.argtypes = [_fp, _fp, _fp, _fp] + [i] * 9
    lib.oracle_conv_int8.argtypes = [_fp, _i8p, _fp, _fp, _i32p] + [i] * 9 + [C.c_float, C.c_float]
    lib.oracle_conv_xnor.argtypes = [_fp, _fp, _fp, _fp, _fp, _i32p] + [i] * 6
    lib.oracle_maxpool.argtypes = [_fp, _fp] + [i] * 9
    lib.oracle_shortcut.argtypes = [_fp, _fp, _fp] + [i] * 8
    lib.oracle_upsample.argtypes = [_fp, _fp] + [i] * 5 + [C.c_float]
    lib.oracle_yolo.argtypes = [_fp, _fp] + [i] * 4
    lib.oracle_region.argtypes = [_fp, _fp] + [i] * 6
    lib.oracle_region_tree.argtypes = [_fp, _fp] + [i] * 5 + [C.POINTER(C.c_int), i]
    lib.oracle_reorg.argtypes = [_fp, _fp] + [i] * 5
    lib.oracle_fuse_bn.argtypes = [_fp, _fp, _fp, _fp, _fp, i, i]
    lib.oracle_binary_mean.argtypes = [_fp, i, i, _fp]
    lib.oracle_binarize.argtypes = [_fp, _fp, C.c_size_t]
    lib.oracle_binarize.restype = None
    lib.oracle_binarize_weights.argtypes = [_fp, _fp, i, i, _fp]
    lib.oracle_binarize_weights.restype = None
    lib.oracle_quantize_weights.argtypes = [_fp, C.c_size_t, _i8p]
    lib.oracle_quantize_weights.restype = C.c_float
    lib.oracle_entropy_calibration.argtypes = [_fp, C.c_size_t, C.c_float, i]
    lib.oracle_entropy_calibration.restype = C.c_float
    lib.oracle_load_resized_u8.argtypes = [C.POINTER(C.c_ubyte)] + [i] * 5 + [_fp]
    lib.oracle_load_resized_u8.restype = None
    lib.oracle_get_boxes.restype = C.c_int
    lib.oracle_get_boxes.argtypes = [C.POINTER(OracleHead), i, i, i, i, i, i, C.c_float, i, i, C.c_float, _fp, i]
    for name in ("oracle_conv_f32", "oracle_conv_int8", "oracle_conv_xnor", "oracle_maxpool", "oracle_shortcut",
                 "oracle_upsample", "oracle_yolo", "oracle_region", "oracle_reorg", "oracle_fuse_bn",
                 "oracle_binary_mean"):
        getattr(lib, name).restype = None
    return lib


ORACLE_FAST_SO = os.path.join(ROOT, "oracle", "liboracle_fast.so")


def oracle_fast_lib() -> C.CDLL:
    """oracle/fast_oracle.c: the INT8 convolution of the oracle with an exact (integer) reordering +
    OpenMP, pinned bit for bit against oracle_conv_int8 (tests/test_oracle_pin.py)."""
    src = os.path.join(ROOT, "oracle", "fast_oracle.c")
    if not (os.path.exists(ORACLE_FAST_SO) and os.path.getmtime(ORACLE_FAST_SO) >= os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"],
                              stdout=subprocess.DEVNULL)
    lib = C.CDLL(ORACLE_FAST_SO)
    lib.oracle_conv_int8_fast.argtypes = [_fp, _i8p, _fp, _fp, _i32p] + [C.c_int] * 9 + [C.c_float, C.c_float]
    lib.oracle_conv_int8_fast.restype = None
    _dp = C.POINTER(C.c_double)
    lib.oracle_conv_f64.argtypes = [_dp, _fp, _fp, _dp] + [C.c_int] * 9
    lib.oracle_conv_f64.restype = None
    return lib


def fp(a: np.ndarray):
    return a.ctypes.data_as(_fp)


# ---------------------------------------------------------------------------
# model fixtures
# ---------------------------------------------------------------------------
_WORK = None


def workdir() -> str:
    global _WORK
    if _WORK is None:
        _WORK = tempfile.mkdtemp(prefix="yl_tests_")
    return _WORK


_MODEL_CACHE = {}


ALL_ACTIVATIONS = ("logistic", "loggy", "relu", "elu", "relie", "plse", "hardtan", "lhtan", "linear", "ramp", "leaky",
                   "tanh", "stair")


def all_activations_cfg(width: int, height: int) -> str:
    """A small trunk that uses every activation of the reference's activate() (src/additionally.h:132-165): in
    3x3 / 1x1 / stride-2 convolutions (direct and Winograd shapes), in [shortcut] layers and in an xnor convolution."""
    c = zoo._Cfg()
    zoo._net(c, width, height, calib=[8.0] * 20)
    c.conv(16, 3, act="leaky")
    for i, act in enumerate(ALL_ACTIVATIONS):
        c.conv(16, 3 if i % 2 == 0 else 1, act=act)
        if i % 4 == 3:
            c.section("shortcut", **{"from": -3, "activation": ALL_ACTIVATIONS[(i * 5) % len(ALL_ACTIVATIONS)]})
    c.conv(32, 3, stride=2, act="elu")
    c.conv(32, 3, act="ramp", xnor=1)           # bit path + a pass of its own for the activation
    c.conv(32, 1, act="tanh", xnor=1)           # xnor FP32 fallback
    c.section("shortcut", **{"from": -2, "activation": "relu"})
    c.conv(18, 1, bn=False, act="logistic")
    return c.text()


def model_files(name: str, width: int, height: int, seed: int = 1):
    """(cfg_path, weights_path) of a generated cfg + synthetic weights."""
    key = (name, width, height, seed)
    if key not in _MODEL_CACHE and name == "all-activations":
        cfg = os.path.join(workdir(), "all-activations-%dx%d.cfg" % (width, height))
        with open(cfg, "w") as f:
            f.write(all_activations_cfg(width, height))
        wpath = cfg[:-4] + "-s%d.weights" % seed
        weights.write_synthetic_weights(open(cfg).read(), wpath, seed=seed)
        _MODEL_CACHE[key] = (cfg, wpath)
    if key not in _MODEL_CACHE:
        cfg = zoo.write_cfg(name, workdir(), width, height)
        wpath = os.path.join(workdir(), "%s-%dx%d-s%d.weights" % (name, width, height, seed))
        with open(cfg) as f:
            weights.write_synthetic_weights(f.read(), wpath, seed=seed)
        _MODEL_CACHE[key] = (cfg, wpath)
    return _MODEL_CACHE[key]


def _variant_default() -> int:
    """YL_VARIANT_DEFAULT as csrc/kernels.h defines it (what yl_network_set_variant(net, -1) selects)"""
    import re
    text = open(os.path.join(ROOT, "yolo2_light_amd", "csrc", "kernels.h")).read()
    m = re.search(r"constexpr int YL_VARIANT_DEFAULT = ([0-9 |]+);", text)
    v = 0
    for tok in m.group(1).split("|"):
        v |= int(tok)
    return v


VARIANT_DEFAULT = _variant_default()


def seeded_input(batch: int, c: int, h: int, w: int, seed: int = 2222222) -> np.ndarray:
    """U[0,1) images, seed echoing srand(2222222) (src/main.c:165)."""
    rng = np.random.default_rng(seed)
    return rng.random((batch, c, h, w), dtype=np.float32)


def write_ppm(path: str, pix: np.ndarray) -> None:
    """binary PPM (P6) of an HWC u8 image -- a lossless container the reference's stb decoder reads"""
    h, w, c = pix.shape
    assert c == 3 and pix.dtype == np.uint8
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(pix).tobytes())


def oracle_load_resized(olib, pix: np.ndarray, w: int, h: int) -> np.ndarray:
    sh, sw, sc = pix.shape
    pix = np.ascontiguousarray(pix)
    out = np.zeros((sc, h, w), dtype=np.float32)
    olib.oracle_load_resized_u8(pix.ctypes.data_as(C.POINTER(C.c_ubyte)), sw, sh, sc, w, h, fp(out))
    return out


def xnor_fallback_operands(olib, x, weights, mean_arr, li):
    """(binarised input, binarised weights) of an xnor conv that is NOT on the bit path."""
    xb = np.empty_like(x)
    olib.oracle_binarize(fp(x), fp(xb), x.size)
    wb = np.empty_like(weights)
    olib.oracle_binarize_weights(fp(weights), fp(mean_arr), li["n"], li["c"] * li["size"] ** 2, fp(wb))
    return xb, wb


# ---------------------------------------------------------------------------
# oracle network walker
# ---------------------------------------------------------------------------
class OracleNet:
    """Runs the C restatement layer by layer over the topology/parameters of a
    host-side `Network` (whose parser and prep passes are pinned separately
    against the reference in tests/test_host_prep.py)."""

    def __init__(self, net: Network, olib: C.CDLL):
        self.net = net
        self.o = olib
        self.infos = net.layers()
        self.outputs = [None] * net.n
        self.int8_acc = {}
        self.xnor_counts = {}
        # route inputs are not exposed through layer_info: recover from the cfg walker in tests
        self.route_inputs = {}

    def set_route_inputs(self, cfg_text: str) -> None:
        secs = zoo.parse_sections(cfg_text)[1:]
        for i, (typ, o) in enumerate(secs):
            if typ == "route":
                ids = [int(x) for x in o["layers"].split(",")]
                self.route_inputs[i] = [j + i if j < 0 else j for j in ids]

    def forward(self, x: np.ndarray, stop_after: int = None) -> None:
        net, o = self.net, self.o
        B = net.batch
        cur = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        for i, li in enumerate(self.infos):
            out = np.zeros(B * li["outputs"], dtype=np.float32)
            t = li["type"]
            if t == CONV:
                wts = net.layer_weights(i)
                bias = net.layer_biases(i)
                mode = li["conv_mode"]
                if mode == CONV_F32:
                    src = cur
                    if li["xnor"]:      # FP32 fallback of an xnor conv: +-mean weights, +-1 input, zero padding
                        src, wts = xnor_fallback_operands(o, cur, wts, net.layer_mean_arr(i), li)
                    o.oracle_conv_f32(fp(src), fp(wts), fp(bias), fp(out), B, li["c"], li["h"], li["w"], li["n"],
                                      li["size"], li["stride"], li["pad"], li["activation"])
                elif mode == CONV_INT8:
                    wq = net.layer_weights_int8(i)
                    im, wm = net.layer_quant_multipliers(i)
                    acc = np.zeros(B * li["outputs"], dtype=np.int32)
                    o.oracle_conv_int8(fp(cur), wq.ctypes.data_as(_i8p), fp(bias), fp(out),
                                       acc.ctypes.data_as(_i32p), B, li["c"], li["h"], li["w"], li["n"],
                                       li["size"], li["stride"], li["pad"], li["activation"], im, wm)
                    self.int8_acc[i] = acc
                else:
                    mean = net.layer_mean_arr(i)
                    cnt = np.zeros(B * li["outputs"], dtype=np.int32)
                    o.oracle_conv_xnor(fp(cur), fp(wts), fp(mean), fp(bias), fp(out), cnt.ctypes.data_as(_i32p),
                                       B, li["c"], li["h"], li["w"], li["n"], li["activation"])
                    self.xnor_counts[i] = cnt
            elif t == MAXPOOL:
                o.oracle_maxpool(fp(cur), fp(out), li["size"], li["w"], li["h"], li["out_w"], li["out_h"], li["c"],
                                 li["pad"], li["stride"], B)
            elif t == ROUTE:
                ids = self.route_inputs[i]
                out2 = out.reshape(B, li["outputs"])
                off = 0
                for j in ids:
                    sz = self.infos[j]["outputs"]
                    out2[:, off:off + sz] = self.outputs[j].reshape(B, sz)
                    off += sz
                out = out2.reshape(-1)
            elif t == SHORTCUT:
                add = self.outputs[li["index"]]
                o.oracle_shortcut(fp(cur), fp(add), fp(out), B, li["w"], li["h"], li["c"], li["out_w"], li["out_h"],
                                  li["out_c"], li["activation"])
            elif t == UPSAMPLE:
                o.oracle_upsample(fp(cur), fp(out), B, li["c"], li["h"], li["w"], li["stride"], 1.0)
            elif t == YOLO:
                o.oracle_yolo(fp(cur), fp(out), B, li["n"], li["classes"], li["w"] * li["h"])
            elif t == REGION:
                tree = self.net.layer_tree(i)
                if tree is not None:
                    gs = np.ascontiguousarray(tree[1], dtype=np.int32)
                    o.oracle_region_tree(fp(cur), fp(out), B, li["n"], li["classes"], li["coords"], li["w"] * li["h"],
                                         gs.ctypes.data_as(C.POINTER(C.c_int)), len(gs))
                else:
                    o.oracle_region(fp(cur), fp(out), B, li["n"], li["classes"], li["coords"], li["w"] * li["h"],
                                    li["softmax"])
            elif t == REORG:
                o.oracle_reorg(fp(cur), fp(out), B, li["out_c"], li["out_h"], li["out_w"], li["stride"])
            else:
                raise NotImplementedError("oracle walker: layer type %d" % t)
            self.outputs[i] = out
            cur = out
            if stop_after is not None and i >= stop_after:
                break


class TruthNet:
    """FLOAT64 ground truth of an FP32 network (yolov3 / yolov3-tiny layer types): the same float weights and
    biases, every tensor and every operation in double (oracle_conv_f64 + numpy).  `outputs[i]` are float64.
    Not a reference path -- the yardstick against which the reference's scalar build, its AVX build and the HIP
    kernels are each measured (tests/test_gpu_parity.py::test_fp32_error_vs_float64_truth)."""

    def __init__(self, net: Network, cfg_text: str):
        self.net = net
        self.f = oracle_fast_lib()
        self.infos = net.layers()
        self.outputs = [None] * net.n
        self.route_inputs = {}
        secs = zoo.parse_sections(cfg_text)[1:]
        for i, (typ, o) in enumerate(secs):
            if typ == "route":
                ids = [int(x) for x in o["layers"].split(",")]
                self.route_inputs[i] = [j + i if j < 0 else j for j in ids]

    @staticmethod
    def _maxpool(x, size, stride, pad):
        B, Cc, H, W = x.shape
        oh, ow = (H + pad - size) // stride + 1, (W + pad - size) // stride + 1
        off = pad // 2                                    # window origin -pad/2 (src/additionally.c:1452)
        xp = np.full((B, Cc, off + (oh - 1) * stride + size + H, off + (ow - 1) * stride + size + W), -np.inf)
        xp[:, :, off:off + H, off:off + W] = x
        out = np.full((B, Cc, oh, ow), -np.inf)
        for ky in range(size):
            for kx in range(size):
                out = np.maximum(out, xp[:, :, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride])
        return out

    def forward(self, x: np.ndarray) -> None:
        net, B = self.net, self.net.batch
        dp = C.POINTER(C.c_double)
        cur = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        for i, li in enumerate(self.infos):
            t = li["type"]
            if t == CONV:
                assert li["conv_mode"] == CONV_F32 and not li["xnor"]
                out = np.zeros(B * li["outputs"], dtype=np.float64)
                wts, bias = net.layer_weights(i), net.layer_biases(i)
                self.f.oracle_conv_f64(cur.ctypes.data_as(dp), fp(wts), fp(bias), out.ctypes.data_as(dp), B, li["c"],
                                       li["h"], li["w"], li["n"], li["size"], li["stride"], li["pad"], li["activation"])
            elif t == MAXPOOL:
                out = self._maxpool(cur.reshape(B, li["c"], li["h"], li["w"]), li["size"], li["stride"], li["pad"])
                out = np.ascontiguousarray(out).reshape(-1)
            elif t == ROUTE:
                parts = [self.outputs[j].reshape(B, -1) for j in self.route_inputs[i]]
                out = np.ascontiguousarray(np.concatenate(parts, axis=1)).reshape(-1)
            elif t == SHORTCUT:
                add = self.outputs[li["index"]]
                assert add.size == cur.size and li["activation"] == 3      # same-shape, linear (every shipped cfg)
                out = cur + add
            elif t == UPSAMPLE:
                v = cur.reshape(B, li["c"], li["h"], li["w"])
                out = np.ascontiguousarray(v.repeat(li["stride"], axis=2).repeat(li["stride"], axis=3)).reshape(-1)
            elif t == YOLO:
                v = cur.reshape(B, li["n"], li["classes"] + 5, li["w"] * li["h"]).copy()
                sel = [0, 1] + list(range(4, li["classes"] + 5))
                v[:, :, sel] = 1.0 / (1.0 + np.exp(-v[:, :, sel]))
                out = v.reshape(-1)
            else:
                raise NotImplementedError("TruthNet: layer type %d" % t)
            self.outputs[i] = out
            cur = out


def error_vs_truth(got: np.ndarray, truth: np.ndarray):
    """(relative RMS error, max |error| / RMS(truth)) of a float32 tensor against the float64 truth"""
    g = np.asarray(got, dtype=np.float64).reshape(-1)
    t = np.asarray(truth, dtype=np.float64).reshape(-1)
    rms = max(float(np.sqrt(np.mean(t * t))), 1e-300)
    e = g - t
    return float(np.sqrt(np.mean(e * e))) / rms, float(np.max(np.abs(e))) / rms


# ---------------------------------------------------------------------------
# comparisons
# ---------------------------------------------------------------------------
FP32_RTOL = 1e-4      # north_star: "within 1e-4 rel for FP32"


def fp32_close(got: np.ndarray, ref: np.ndarray, rtol: float = FP32_RTOL):
    """|got-ref| <= rtol*|ref| + rtol*RMS(ref): relative tolerance with an absolute
    floor tied to the layer's own scale for cancellation-limited elements.
    Returns (ok, max_err_over_allowed, worst_index)."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    rms = float(np.sqrt(np.mean(ref * ref))) if ref.size else 0.0
    allowed = rtol * np.abs(ref) + rtol * max(rms, 1e-30)
    err = np.abs(got - ref)
    ratio = err / allowed
    worst = int(np.argmax(ratio)) if ratio.size else 0
    return bool(np.all(np.isfinite(got)) and (ratio.size == 0 or ratio[worst] <= 1.0)), \
        float(ratio[worst]) if ratio.size else 0.0, worst


def strict_max_rel(got: np.ndarray, ref: np.ndarray, floor_frac: float = 0.01) -> float:
    """max |got-ref| / |ref| over the elements with |ref| > floor_frac * RMS(ref): the PURE relative error, without
    fp32_close's RMS-tied absolute allowance (report it next to fp32_close).  Elements below 1 % of
    the layer RMS are cancellation results whose relative error is set by the summation order, not by the kernel."""
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    if not ref.size:
        return 0.0
    rms = float(np.sqrt(np.mean(ref * ref)))
    sel = np.abs(ref) > floor_frac * max(rms, 1e-30)
    if not sel.any():
        return 0.0
    return float(np.max(np.abs(got[sel] - ref[sel]) / np.abs(ref[sel])))


class OracleHeads:
    """The detection heads of a network as the oracle's decode wants them.  `outputs[i]` = the head
    tensor of layer i ([batch*outputs] float32): by default downloaded from the device network."""

    def __init__(self, net, outputs: dict = None):
        self.netw, self.neth, _ = net.input_dims
        self.keep = []
        heads = []
        for i, li in enumerate(net.layers()):
            if li["type"] not in (YOLO, REGION):
                continue
            mask, anchors = net.layer_head(i)
            out = np.ascontiguousarray(outputs[i] if outputs is not None else net.layer_output(i), dtype=np.float32)
            mask = np.ascontiguousarray(mask, dtype=np.int32)
            anchors = np.ascontiguousarray(anchors, dtype=np.float32)
            self.keep += [out, mask, anchors]
            tree = net.layer_tree(i) if li["type"] == REGION else None
            tparent = None
            if tree is not None:
                tp = np.ascontiguousarray(tree[0], dtype=np.int32)
                self.keep.append(tp)
                tparent = tp.ctypes.data_as(C.POINTER(C.c_int))
            heads.append(OracleHead(li["type"], li["w"], li["h"], li["n"], li["classes"], li["outputs"], fp(out),
                                    mask.ctypes.data_as(C.POINTER(C.c_int)), fp(anchors), tparent))
        self.classes = heads[-1].classes
        self.arr = (OracleHead * len(heads))(*heads)
        self.n = len(heads)


_OLIB = None


def oracle_boxes(net_or_heads, image: int, w: int, h: int, thresh: float, nms: float = 0.0, relative: int = 1,
                 letter: int = 0, max_rows: int = 200000) -> np.ndarray:
    """get_network_boxes + do_nms_sort of batch item `image` by the oracle (oracle/detect_oracle.c, pinned
    row for row against the reference in tests/test_detect_host.py) on the network's head tensors."""
    global _OLIB
    if _OLIB is None:
        _OLIB = oracle_lib()
    hd = net_or_heads if isinstance(net_or_heads, OracleHeads) else OracleHeads(net_or_heads)
    rows = np.zeros((max_rows, 6 + hd.classes), dtype=np.float32)
    n = _OLIB.oracle_get_boxes(hd.arr, hd.n, hd.netw, hd.neth, image, w, h, thresh, relative, letter, nms,
                               fp(rows), max_rows)
    assert n >= 0
    return rows[:min(n, max_rows)].copy()


def require_ref(fast: bool = False, hip: bool = False) -> None:
    """GPU tests that compare against the reference-built libraries (oracle/_ref, built in the container where
    /root/reference exists and shipped with the gpurun snapshot): on a GPU box a missing library is a FAILURE -- the
    parity gate must not silently turn into a skip -- and a skip only where there is no GPU at all."""
    import pytest
    missing = []
    if not refbind.available():
        missing.append(refbind.GOLD)
    if fast and not refbind.available(fast=True):
        missing.append(refbind.FAST)
    if hip and not os.path.exists(refbind.HIP):
        missing.append(refbind.HIP)
    if not missing:
        return
    msg = "reference-built libraries missing: %s (run `make -C oracle` where /root/reference exists)" % ", ".join(missing)
    if have_gpu():
        pytest.fail(msg)
    pytest.skip(msg)


def have_gpu() -> bool:
    from yolo2_light_amd._lib import lib
    return lib.yl_device_count() > 0
