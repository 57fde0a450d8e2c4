"""GPU parity for the INT8 (K2) and XNOR (K3) paths and a teacher-forced
layer-by-layer check of whole networks in all three modes.

Bars: XNOR match counts and INT8 int16-clamped accumulators bit-exact; their
FP32 epilogues replay the reference's scalar float ops, so the layer outputs
are bit-exact too.  "Teacher forcing" = every layer is checked against the
oracle applied to the GPU's OWN input of that layer, so one rounding-level
difference upstream (e.g. a sign flip of a near-zero activation) cannot mask or
fake a kernel bug downstream.
"""
import ctypes as C
import os

import numpy as np
import pytest

import common
import descs as D
from common import Network, fp, fp32_close
from yolo2_light_amd._lib import check, lib

pytestmark = pytest.mark.gpu

_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)


def _net_from(descs_list, batch, w, h, c, quantized=0):
    net = Network.from_desc(descs_list, batch, w, h, c, quantized)
    check(lib.yl_network_set_debug(net._h, 1), "set_debug")
    net.to_device(0)
    return net


XNOR_SHAPES = [
    # B, C, H, W, M
    (2, 16, 13, 17, 32),      # c%32 != 0 branch of the reference, Cw = 1
    (1, 32, 26, 26, 64),
    (3, 64, 7, 9, 40),        # M not a multiple of the filter tile
    (1, 100, 12, 10, 33),     # ragged channels: Cw = 2 with 28 pad bits
    (2, 256, 13, 13, 70),     # Cw = 4
    (1, 512, 6, 5, 130),      # Cw = 8
    (5, 192, 3, 3, 16),       # Cw = 3 (odd) and many tiny images per block
]


@pytest.mark.parametrize("variant", [30, 30 | 512], ids=["ft-by-grid", "ft64"])
@pytest.mark.parametrize("shape", XNOR_SHAPES)
def test_conv_xnor_bit_exact(olib, shape, variant):
    """variant bit 9: 64-filter workgroups wherever the layer has 64 filters (at these sizes the default picks 32)"""
    B, Cc, H, W, M = shape
    rng = np.random.default_rng(77 + Cc + M)
    K = Cc * 9
    wts = rng.normal(0, 1.0, M * K).astype(np.float32)
    wts[rng.random(M * K) < 0.02] = 0.0                      # exact zeros: bit must be 0 (w > 0 is false)
    bias = rng.normal(0, 0.5, M).astype(np.float32)
    mean = np.zeros(M, np.float32)
    olib.oracle_binary_mean(fp(wts), M, K, fp(mean))
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    x[rng.random(x.shape) < 0.05] = 0.0                       # x == 0 -> bit 0
    d = D.conv(B, W, H, Cc, M, 3, 1, 1, D.LEAKY, wts, bias, xnor=1, mean_arr=mean)
    net = _net_from([d], B, W, H, Cc)
    net.set_variant(variant)
    got = net.predict(x)
    cnt = net.layer_xnor_counts(0)
    ref = np.zeros_like(got)
    rcnt = np.zeros(got.size, np.int32)
    olib.oracle_conv_xnor(fp(x), fp(wts), fp(mean), fp(bias), fp(ref), rcnt.ctypes.data_as(_i32p), B, Cc, H, W, M, D.LEAKY)
    assert np.array_equal(cnt, rcnt), "match counts differ: %d of %d" % (np.count_nonzero(cnt != rcnt), cnt.size)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    net.close()


INT8_SHAPES = [
    # B, C, H, W, M, size, stride, pad
    (2, 16, 13, 13, 32, 3, 1, 1),
    (1, 32, 19, 23, 64, 3, 2, 1),
    (2, 64, 9, 7, 255, 1, 1, 0),
    (1, 100, 8, 8, 24, 3, 1, 1),       # channels padded 100 -> 128 (G = 8)
    (3, 128, 13, 13, 130, 1, 1, 0),
    (1, 256, 10, 10, 96, 3, 1, 1),
    (4, 48, 5, 5, 40, 3, 1, 1),        # G = 3 -> padded to 4
]


INT8_SHAPES += [
    (2, 128, 12, 12, 128, 3, 1, 1),    # M % 128 == 0: the wide tiles without row masks
    (1, 256, 19, 19, 256, 3, 1, 1),    # odd map (yolov3-608's last scale), two 128-row tiles, 18 panels
    (3, 64, 8, 8, 64, 1, 1, 0),        # one panel only (nkb = 1)
]


@pytest.mark.parametrize("shape", INT8_SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
def test_conv_int8_bit_exact(olib, shape, tile):
    B, Cc, H, W, M, size, stride, pad = shape
    rng = np.random.default_rng(99 + Cc + M)
    K = Cc * size * size
    wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
    bias = rng.normal(0, 0.5, M).astype(np.float32)
    wq = np.zeros(M * K, np.int8)
    w_mult = olib.oracle_quantize_weights(fp(wts), M * K, wq.ctypes.data_as(_i8p))
    in_mult = 15.497
    x = (rng.standard_normal((B, Cc, H, W)) * 3).astype(np.float32)
    x.reshape(-1)[:7] = [1e9, -1e9, 40000.7 / in_mult, -33000.2 / in_mult, 127.9 / in_mult, -128.5 / in_mult, 0.0]
    l0 = D.maxpool(B, W, H, Cc, 1, 1, pad=0)                  # identity: INT8 never runs on layer 0
    l1 = D.conv(B, W, H, Cc, M, size, stride, pad, D.LEAKY, wts, bias, weights_int8=wq, in_mult=in_mult, w_mult=w_mult)
    net = _net_from([l0, l1], B, W, H, Cc, quantized=1)
    assert net.layer_info(1)["int8"] == 1
    net.set_int8_tile(tile)
    got = net.predict(x)
    acc = net.layer_int8_acc(1)
    ref = np.zeros_like(got)
    racc = np.zeros(got.size, np.int32)
    olib.oracle_conv_int8(fp(x), wq.ctypes.data_as(_i8p), fp(bias), fp(ref), racc.ctypes.data_as(_i32p),
                          B, Cc, H, W, M, size, stride, pad, D.LEAKY, in_mult, w_mult)
    assert np.array_equal(acc, racc), "int16 accumulators differ: %d of %d" % (np.count_nonzero(acc != racc), acc.size)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    net.close()


# ----------------------------------------------------------------------------
# teacher-forced whole networks
# ----------------------------------------------------------------------------
def _teacher_forced(olib, name, width, height, batch, quantized, x=None):
    cfg, wts = common.model_files(name, width, height)
    net = Network.load(cfg, wts, batch, quantized, device=0, debug=True)
    if x is None:
        x = common.seeded_input(batch, 3, height, width)
    x = np.ascontiguousarray(x, dtype=np.float32)
    net.predict(x)
    infos = net.layers()
    outs = [net.layer_output(i) for i in range(net.n)]
    route_inputs = {}
    from yolo2_light_amd import zoo
    for i, (typ, o) in enumerate(zoo.parse_sections(open(cfg).read())[1:]):
        if typ == "route":
            ids = [int(v) for v in o["layers"].split(",")]
            route_inputs[i] = [j + i if j < 0 else j for j in ids]
    B = batch
    stats = {"exact": 0, "fp32": 0}
    for i, li in enumerate(infos):
        cur = x.reshape(-1) if i == 0 else outs[i - 1]
        ref = np.zeros(B * li["outputs"], np.float32)
        t = li["type"]
        exact = True
        if t == common.CONV:
            w_ = net.layer_weights(i); b_ = net.layer_biases(i)
            if li["conv_mode"] == common.CONV_F32:
                if li["xnor"]:
                    cur, w_ = common.xnor_fallback_operands(olib, cur, w_, net.layer_mean_arr(i), li)
                olib.oracle_conv_f32(fp(cur), fp(w_), fp(b_), fp(ref), B, li["c"], li["h"], li["w"], li["n"],
                                     li["size"], li["stride"], li["pad"], li["activation"])
                exact = False
            elif li["conv_mode"] == common.CONV_INT8:
                wq = net.layer_weights_int8(i)
                im, wm = net.layer_quant_multipliers(i)
                racc = np.zeros(ref.size, np.int32)
                olib.oracle_conv_int8(fp(cur), wq.ctypes.data_as(_i8p), fp(b_), fp(ref), racc.ctypes.data_as(_i32p),
                                      B, li["c"], li["h"], li["w"], li["n"], li["size"], li["stride"], li["pad"],
                                      li["activation"], im, wm)
                assert np.array_equal(net.layer_int8_acc(i), racc), "layer %d int8 accumulators" % i
            else:
                mean = net.layer_mean_arr(i)
                rcnt = np.zeros(ref.size, np.int32)
                olib.oracle_conv_xnor(fp(cur), fp(w_), fp(mean), fp(b_), fp(ref), rcnt.ctypes.data_as(_i32p),
                                      B, li["c"], li["h"], li["w"], li["n"], li["activation"])
                assert np.array_equal(net.layer_xnor_counts(i), rcnt), "layer %d xnor counts" % i
        elif t == common.MAXPOOL:
            olib.oracle_maxpool(fp(cur), fp(ref), li["size"], li["w"], li["h"], li["out_w"], li["out_h"], li["c"],
                                li["pad"], li["stride"], B)
        elif t == common.ROUTE:
            ref = np.concatenate([outs[j].reshape(B, -1) for j in route_inputs[i]], axis=1).reshape(-1)
        elif t == common.SHORTCUT:
            olib.oracle_shortcut(fp(cur), fp(outs[li["index"]]), fp(ref), B, li["w"], li["h"], li["c"],
                                 li["out_w"], li["out_h"], li["out_c"], li["activation"])
        elif t == common.UPSAMPLE:
            olib.oracle_upsample(fp(cur), fp(ref), B, li["c"], li["h"], li["w"], li["stride"], 1.0)
        elif t == common.YOLO:
            olib.oracle_yolo(fp(cur), fp(ref), B, li["n"], li["classes"], li["w"] * li["h"])
            exact = False
        elif t == common.REGION:
            olib.oracle_region(fp(cur), fp(ref), B, li["n"], li["classes"], li["coords"], li["w"] * li["h"], li["softmax"])
            exact = False
        elif t == common.REORG:
            olib.oracle_reorg(fp(cur), fp(ref), B, li["out_c"], li["out_h"], li["out_w"], li["stride"])
        else:
            raise AssertionError("unexpected layer type %d" % t)
        got = outs[i]
        if exact:
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "layer %d %r not bit-exact" % (i, li)
            stats["exact"] += 1
        else:
            ok, ratio, worst = fp32_close(got, ref)
            assert ok, "layer %d %r: err/allowed %.3g at %d (got %r ref %r)" % (i, li, ratio, worst, got[worst], ref[worst])
            stats["fp32"] += 1
    net.close()
    return stats


@pytest.mark.parametrize("name,width,height,batch,quantized", [
    ("tiny-yolo-xnor", 96, 96, 2, 0),
    ("tiny-yolo-xnor", 416, 416, 1, 0),         # BASELINE config 5 resolution
    ("yolov3-tiny", 96, 96, 2, 1),
    ("yolov3", 64, 64, 2, 1),
    ("yolov3", 96, 64, 1, 0),
])
def test_network_teacher_forced(olib, name, width, height, batch, quantized):
    stats = _teacher_forced(olib, name, width, height, batch, quantized)
    assert stats["exact"] > 0


def test_int8_network_vs_reference_library_batch1():
    """-quantized yolov3-tiny 416 END TO END against network_predict_quantized of the reference itself (which handles batch
    item 0 only) -- pinned where it is measured (profiles/r6_int8_end_to_end_diag.txt).

    Quantisation is a step function of the FP32 first layer's output.  That layer here is gemm_nn's k-order chain of FUSED
    multiply-adds; the reference's `make AVX=1` build runs the same chain (_mm256_fmadd_ps, src/additionally.c gemm_nn), its
    scalar build rounds every product and every sum separately: against the scalar build 58 % of layer 0's outputs differ by
    <= 1 ulp (7e-7), 66 of the 1.38 M int8 codes of the next layer flip, and the synthetic i.i.d. weights amplify that ~1.4 x
    per layer to 5.5e-3 at the heads (1.3e-2 at yolov3-608).  So:
      * against the reference's AVX build (bit-equal first layer, no flips): the heads agree to FP32 rounding of the linear
        head convolutions -- asserted at 1e-4 relative RMS; a wrong dequantisation scale, clamp or /32 on any layer is orders
        of magnitude above that;
      * against the scalar build: the flip fraction at the first INT8 layer is asserted (< 2e-4 of its outputs differ at all),
        the heads at 5e-2 -- the looser bound is the amplification of those few flips, not slack in a kernel."""
    common.require_ref(fast=True)
    name, width, height = "yolov3-tiny", 416, 416
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(1, 3, height, width)
    net = Network.load(cfg, wts, 1, 1, device=0)
    net.predict(x)
    first_i8 = next(i for i in range(net.n) if net.layer_kernel(i).startswith("conv_i8"))
    for fast, bound in ((True, 1e-4), (False, 5e-2)):
        ref = common.refbind.RefNetwork(cfg, wts, 1, 1, fast=fast)
        ref.predict(x)
        for i in range(net.n):
            li = net.layer_info(i)
            if li["type"] not in (common.YOLO,):
                continue
            g = net.layer_output(i).astype(np.float64); r = ref.layer_output(i).astype(np.float64)
            rms = np.sqrt(np.mean(r * r))
            rel_rms_err = np.sqrt(np.mean((g - r) ** 2)) / rms
            print("yolo layer %d: INT8 end-to-end relative RMS error vs the reference's %s build %.3g" % (
                i, "AVX" if fast else "scalar", rel_rms_err))
            assert rel_rms_err < bound, "yolo layer %d vs the %s build: relative RMS error %.3g" % (i, "AVX" if fast else "scalar", rel_rms_err)
        # an int8 code that flipped moves an output by one quantisation step of one input (>= 1e-3 of the layer's RMS); FP32 rounding
        # of the dequantise / bias / leaky tail (the AVX build runs it under -Ofast) stays far below that
        g2, r2 = net.layer_output(first_i8).astype(np.float64), ref.layer_output(first_i8).astype(np.float64)
        flips = float(np.mean(np.abs(g2 - r2) > 1e-3 * np.sqrt(np.mean(r2 * r2))))
        print("first INT8 layer (%d): %.3g of its outputs differ from the %s build by more than 1e-3 of the layer RMS" % (
            first_i8, flips, "AVX" if fast else "scalar"))
        assert flips < (1e-6 if fast else 2e-4)
        if not fast:
            r = ref.get_detections(0, width, height, 0.24, nms=0.4)
            g = net.get_boxes(0, width, height, 0.24, nms=0.4)
            assert abs(len(r) - len(g)) <= max(2, len(r) // 50)
    net.close()


@pytest.mark.parametrize("width,height,batch,tile", [(96, 96, 2, 0), (96, 96, 2, 1), (96, 96, 2, 3), (96, 96, 2, 4), (96, 96, 2, 6),
                                                     (96, 96, 2, 5), (160, 96, 3, 0), (160, 96, 3, 3), (224, 160, 1, 4)])
def test_int8_fusion_is_bit_identical(width, height, batch, tile):
    """-quantized yolov3 with yl_network_set_fusion: conv+[shortcut] folded, the next layer's int8
    input written from the producer's epilogue (no separate quantise pass), unread FP32 tensors
    skipped.  Every tensor that is still materialised must equal the unfused run bit for bit -- for every
    tile configuration of the INT8 kernel (odd maps down to 3x3: tiles span several images)."""
    name = "yolov3"
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 1, device=0)
    fused = Network.load(cfg, wts, batch, 1, device=0, fuse=True)
    fused.set_int8_tile(tile)
    plain.predict(x)
    fused.predict(x)
    infos = plain.layers()
    checked = 0
    for i, li in enumerate(infos):
        if (li["type"] in (common.SHORTCUT, common.ROUTE, common.YOLO, common.UPSAMPLE) or
                (li["type"] == common.CONV and li["activation"] == D.LINEAR)) and fused.layer_materialised(i):
            a, b = plain.layer_output(i), fused.layer_output(i)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "layer %d %r" % (i, li)
            checked += 1
    assert checked >= 26          # 23 shortcuts + 3 heads; the two multi-input routes are quantised source by source, the upsampled tensors in front of them never written
    for b in range(batch):
        assert np.array_equal(plain.get_boxes(b, width, height, 0.24, nms=0.4),
                              fused.get_boxes(b, width, height, 0.24, nms=0.4))
    plain.close(); fused.close()


@pytest.mark.parametrize("variant", [8, 8 | 16384], ids=["valu", "mfma"])
@pytest.mark.parametrize("width,height,batch", [(96, 96, 2), (160, 96, 3)])
def test_int8_fusion_with_first_layer_kernel_is_bit_identical(width, height, batch, variant):
    """-quantized yolov3, fused, layer 0 through the first-layer kernels (conv_f32_first.hip on the VALU, variant bit 3;
    conv_f32_firstm.hip on the FP32 matrix pipe, + bit 14) writing ONLY the int8 input of layer 1:
    same materialised tensors and detections as the unfused run on the generic kernels."""
    cfg, wts = common.model_files("yolov3", width, height)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 1, device=0)
    fused = Network.load(cfg, wts, batch, 1, device=0, fuse=True)
    plain.set_variant(0)
    fused.set_variant(variant)
    plain.predict(x)
    fused.predict(x)
    assert "conv_f32_first" in fused.layer_kernel(0) and "qonly" in fused.layer_kernel(0), fused.layer_kernel(0)
    assert ("mfma32x32x2" in fused.layer_kernel(0)) == bool(variant & 16384), fused.layer_kernel(0)
    infos = plain.layers()
    checked = 0
    for i, li in enumerate(infos):
        if (li["type"] in (common.SHORTCUT, common.ROUTE, common.YOLO, common.UPSAMPLE) or
                (li["type"] == common.CONV and li["activation"] == D.LINEAR)) and fused.layer_materialised(i):
            assert np.array_equal(plain.layer_output(i).view(np.uint32), fused.layer_output(i).view(np.uint32)), i
            checked += 1
    assert sum(1 for i, li in enumerate(infos) if li["type"] == common.UPSAMPLE and not fused.layer_materialised(i)) == 2
    assert checked >= 26          # 23 shortcuts + 3 heads; the two multi-input routes are quantised source by source, the upsampled tensors in front of them never written
    for b in range(batch):
        assert np.array_equal(plain.get_boxes(b, width, height, 0.24, nms=0.4),
                              fused.get_boxes(b, width, height, 0.24, nms=0.4))
    plain.close(); fused.close()


@pytest.mark.parametrize("width,height,batch", [(40, 24, 3), (104, 88, 1), (32, 16, 2)])
def test_int8_first_layer_units_mfma_equals_valu(olib, width, height, batch):
    """conv(FP32, RGB -> 32, leaky) -> conv(INT8): with fusion on, layer 0 writes only the int8 units of layer 1 -- on the VALU
    (conv_f32_first.hip) or on the FP32 matrix pipe (conv_f32_firstm.hip, variant bit 14; ragged 32 x 16 patches at 40 x 24 and
    104 x 88) -- and both equal the unfused run bit for bit, the `int16_t = float` wrap corner of the quantiser included."""
    rng = np.random.default_rng(width + height)
    M0, M1 = 32, 48
    w0 = rng.normal(0, np.sqrt(2.0 / 27), M0 * 27).astype(np.float32)
    b0 = rng.normal(0, 0.5, M0).astype(np.float32)
    w1 = rng.normal(0, np.sqrt(2.0 / (M0 * 9)), M1 * M0 * 9).astype(np.float32)
    b1 = rng.normal(0, 0.5, M1).astype(np.float32)
    wq = np.zeros(w1.size, np.int8)
    w_mult = olib.oracle_quantize_weights(fp(w1), w1.size, wq.ctypes.data_as(_i8p))
    in_mult = 15.497
    x = (rng.standard_normal((batch, 3, height, width)) * 2).astype(np.float32)
    x[0, 0, 3, 5] = 1e9; x[0, 1, 9, 30] = -1e9; x[-1, 2, height - 1, width - 1] = 3000.0      # |y * mult| >= 32768 somewhere

    def build(fuse, variant):
        l0 = D.conv(batch, width, height, 3, M0, 3, 1, 1, D.LEAKY, w0, b0)
        l1 = D.conv(batch, width, height, M0, M1, 3, 1, 1, D.LEAKY, w1, b1, weights_int8=wq, in_mult=in_mult, w_mult=w_mult)
        net = Network.from_desc([l0, l1], batch, width, height, 3, 1)
        net.set_fusion(fuse)
        net.set_variant(variant)
        net.to_device(0)
        return net

    plain, valu, mfma = build(False, 0), build(True, 8), build(True, 8 | 16384)
    ref = plain.predict(x).copy()
    a, b = valu.predict(x).copy(), mfma.predict(x).copy()
    assert "valu" in valu.layer_kernel(0) and "qonly" in valu.layer_kernel(0), valu.layer_kernel(0)
    assert "mfma32x32x2" in mfma.layer_kernel(0) and "qonly" in mfma.layer_kernel(0), mfma.layer_kernel(0)
    assert np.array_equal(ref.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(ref.view(np.uint32), b.view(np.uint32))
    plain.close(); valu.close(); mfma.close()


@pytest.mark.parametrize("name,width,height,batch,quantized", [
    ("yolov2-voc", 96, 96, 2, 0), ("yolov2-voc", 160, 160, 1, 1), ("tiny-yolo-voc", 96, 64, 2, 0),
    ("yolov3-spp", 64, 64, 2, 0), ("yolov3-spp", 96, 96, 1, 1),
    ("all-activations", 48, 32, 2, 0), ("all-activations", 64, 48, 1, 1),
])
def test_network_teacher_forced_other_cfgs(olib, name, width, height, batch, quantized):
    """The remaining cfgs of the reference's bin/: reorg, region+softmax, SPP max-pools."""
    stats = _teacher_forced(olib, name, width, height, batch, quantized)
    assert stats["exact"] > 0


XNOR_MIXED_CFG = """[net]
batch=1
subdivisions=1
width=%d
height=%d
channels=3
[convolutional]
batch_normalize=1
filters=32
size=3
stride=1
pad=1
activation=leaky
[convolutional]
xnor=1
batch_normalize=1
filters=64
size=3
stride=2
pad=1
activation=leaky
[convolutional]
xnor=1
batch_normalize=1
filters=32
size=1
stride=1
pad=1
activation=leaky
[convolutional]
xnor=1
batch_normalize=1
filters=64
size=3
stride=1
pad=1
activation=leaky
[shortcut]
from=-3
activation=linear
[convolutional]
size=1
stride=1
pad=1
filters=33
activation=linear
[region]
anchors = 1,1, 2,2, 3,3
classes=6
coords=4
num=3
softmax=1
"""


def _mixed_xnor_files(width, height):
    import os
    from yolo2_light_amd import weights as W
    text = XNOR_MIXED_CFG % (width, height)
    cfg = os.path.join(common.workdir(), "xnor-mixed-%dx%d.cfg" % (width, height))
    open(cfg, "w").write(text)
    wts = cfg[:-4] + ".weights"
    W.write_synthetic_weights(text, wts, seed=3)
    return cfg, wts


def test_xnor_fallback_layers_teacher_forced(olib):
    """xnor convs that are NOT 3x3/stride-1 (here 3x3/2 and 1x1) take the reference's FP32
    fallback on binarised operands; the 3x3/1 one takes the bit path; then a shortcut."""
    width, height, batch = 64, 48, 2
    cfg, wts = _mixed_xnor_files(width, height)
    common._MODEL_CACHE[("xnor-mixed", width, height, 1)] = (cfg, wts)
    stats = _teacher_forced(olib, "xnor-mixed", width, height, batch, 0)
    assert stats["exact"] >= 2


@pytest.mark.parametrize("variant", [-1, common.VARIANT_DEFAULT & ~16384], ids=["first-mfma", "first-valu"])
@pytest.mark.parametrize("name,width,height,batch", [("tiny-yolo-xnor", 416, 416, 2), ("tiny-yolo-xnor", 96, 96, 3),
                                                     ("tiny-yolo-xnor", 160, 224, 1), ("tiny-yolo-xnor", 224, 160, 2)])
def test_xnor_sign_domain_fusion_is_bit_identical(name, width, height, batch, variant):
    """yl_network_set_fusion on an XNOR network: conv(xnor) -> [maxpool] -> conv(xnor) chains hand over sign
    words (the producer's epilogue packs (y > 0), max-pooling is the OR of the window), FP32 tensors nobody else
    reads are not written.  Every tensor that is still materialised -- the head above all -- equals the unfused
    run bit for bit."""
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 0, device=0)
    fused = Network.load(cfg, wts, batch, 0, device=0, fuse=True, variant=variant)
    plain.predict(x)
    fused.predict(x)
    # the first layer hands over sign words: on the FP32 matrix pipe by default (conv_f32_firstm.hip), on the VALU without bit 14
    assert ("mfma16x16x4" in fused.layer_kernel(0)) == (variant == -1), fused.layer_kernel(0)
    skipped = checked = 0
    for i in range(plain.n):
        if not fused.layer_materialised(i):
            skipped += 1
            continue
        a, b = plain.layer_output(i), fused.layer_output(i)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "layer %d %r" % (i, plain.layer_info(i))
        checked += 1
    assert skipped >= 12 and checked >= 3          # 6 xnor convs + 6 max-pools + the FP32 first layer and its pool stop writing FP32
    assert not fused.layer_materialised(0) and "conv_f32_first" in fused.layer_kernel(0)      # layer 0 hands over sign words
    for b in range(batch):
        assert np.array_equal(plain.get_boxes(b, width, height, 0.05, nms=0.4), fused.get_boxes(b, width, height, 0.05, nms=0.4))
    plain.close(); fused.close()


XNOR_RAGGED_CFG = """[net]
batch=1
subdivisions=1
width=%d
height=%d
channels=3
[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky
""" + "".join("""[convolutional]
xnor=1
batch_normalize=1
filters=%d
size=3
stride=1
pad=1
activation=leaky
""" % m for m in (70, 130, 33, 100, 64)) + """[convolutional]
size=1
stride=1
pad=1
filters=33
activation=linear
[region]
anchors = 1,1, 2,2, 3,3
classes=6
coords=4
num=3
softmax=1
"""


XNOR_FIRST_POOL_CFG = """[net]
batch=1
subdivisions=1
width=%d
height=%d
channels=3
[convolutional]
batch_normalize=1
filters=%d
size=3
stride=1
pad=1
activation=leaky
[maxpool]
size=2
stride=2
[convolutional]
xnor=1
batch_normalize=1
filters=48
size=3
stride=1
pad=1
activation=leaky
[convolutional]
size=1
stride=1
pad=1
filters=33
activation=linear
[region]
anchors = 1,1, 2,2, 3,3
classes=6
coords=4
num=3
softmax=1
"""


@pytest.mark.parametrize("width,height,filters", [(40, 24, 16), (40, 27, 16), (104, 88, 11), (32, 16, 16)])
def test_xnor_first_layer_sign_words_mfma_equals_valu(width, height, filters):
    """conv(FP32, RGB -> <= 16) -> [maxpool 2x2/2] -> conv(xnor): the first layer hands over sign words -- on the FP32 matrix pipe
    (conv_f32_firstm.hip) with the pooling folded in where the windows are whole (even H and W), without it where they are not (H = 27:
    the OR-pooling kernel follows), on the VALU without variant bit 14 -- and every form equals the unfused run bit for bit; ragged
    32 x 16 patches, 11 filters."""
    from yolo2_light_amd import weights as W
    batch = 3
    text = XNOR_FIRST_POOL_CFG % (width, height, filters)
    cfg = os.path.join(common.workdir(), "xnor-first-pool-%dx%d-%d.cfg" % (width, height, filters))
    open(cfg, "w").write(text)
    wts = cfg[:-4] + ".weights"
    W.write_synthetic_weights(text, wts, seed=21)
    x = common.seeded_input(batch, 3, height, width) - 0.35
    plain = Network.load(cfg, wts, batch, 0, device=0)
    mfma = Network.load(cfg, wts, batch, 0, device=0, fuse=True)
    valu = Network.load(cfg, wts, batch, 0, device=0, fuse=True, variant=common.VARIANT_DEFAULT & ~16384)
    ref = plain.predict(x).copy()
    a, b = mfma.predict(x).copy(), valu.predict(x).copy()
    whole = (height % 2 == 0 and width % 2 == 0)
    assert "mfma16x16x4" in mfma.layer_kernel(0) and ("pool" in mfma.layer_kernel(0)) == whole, mfma.layer_kernel(0)
    assert "valu" in valu.layer_kernel(0) and "signs" in valu.layer_kernel(0), valu.layer_kernel(0)
    assert not mfma.layer_materialised(0) and not mfma.layer_materialised(1)
    assert np.array_equal(ref.view(np.uint32), a.view(np.uint32))
    assert np.array_equal(ref.view(np.uint32), b.view(np.uint32))
    plain.close(); mfma.close(); valu.close()


@pytest.mark.parametrize("variant", [0, 512, 256], ids=["ft-by-grid", "ft64", "float-epilogue"])
def test_xnor_sign_domain_ragged_filter_counts(variant):
    """Sign words between XNOR layers whose filter counts are NOT multiples of 64 (70, 130, 33, 100): the bits above
    the last filter of the last word must be zeros whichever workgroup shape wrote the word (32-filter tiles write
    half words) -- the next layer counts them against pad weight bits of 1.  Fused == unfused, bit for bit."""
    import os
    from yolo2_light_amd import weights as W
    width, height, batch = 40, 24, 3
    text = XNOR_RAGGED_CFG % (width, height)
    cfg = os.path.join(common.workdir(), "xnor-ragged-%dx%d.cfg" % (width, height))
    open(cfg, "w").write(text)
    wts = cfg[:-4] + ".weights"
    W.write_synthetic_weights(text, wts, seed=9)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 0, device=0)
    fused = Network.load(cfg, wts, batch, 0, device=0, fuse=True)
    fused.set_variant(common.VARIANT_DEFAULT | variant)      # the XNOR switches on top of the default kernel selection
    # poison the sign-word ring first: a run of the same network on another image leaves stale words in every slot
    fused.predict(common.seeded_input(batch, 3, height, width, seed=5))
    a = plain.predict(x).copy()
    b = fused.predict(x).copy()
    assert sum(0 if fused.layer_materialised(i) else 1 for i in range(fused.n)) >= 4
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    plain.close(); fused.close()


def test_xnor_sign_thresholds_match_the_float_epilogue():
    """Sign-only XNOR layers compare the match count with a per-filter threshold (conv_xnor.hip).  The thresholds on
    the device must be exactly where fl(fl((2*count - K) * mean) + bias) > 0 switches on, for every filter of every
    XNOR layer; and the run with thresholds equals the run with the float epilogue (variant bit 8) bit for bit."""
    cfg, wts = common.model_files("tiny-yolo-xnor", 96, 96)
    net = Network.load(cfg, wts, 2, 0, device=0, fuse=True)
    seen = 0
    for i, li in enumerate(net.layers()):
        raw = net.layer_packed(i, 4) if li["type"] == common.CONV else None
        if raw is None:
            continue
        thr = raw.view(np.int32)
        mean = net.layer_packed(i, 5).view(np.float32)
        bias = net.layer_packed(i, 6).view(np.float32)
        M, K = mean.size, 9 * li["c"]
        assert thr[-1] == 0                                                   # every filter is a step function
        assert np.all(thr[M:-1] == 0x7fffffff)                                # pad filters never set a bit
        c = np.arange(K + 1, dtype=np.int64)
        v = ((2 * c - K).astype(np.float32)[None, :] * mean[:, None]).astype(np.float32) + bias[:, None]
        pos = v.astype(np.float32) > 0
        want = np.where(pos.any(axis=1), pos.argmax(axis=1), K + 1)
        assert np.array_equal(thr[:M], want.astype(np.int32)), i
        assert np.all(pos == (c[None, :] >= want[:, None]))
        seen += 1
    assert seen >= 6
    x = common.seeded_input(2, 3, 96, 96)
    a = net.predict(x).copy()
    for bits in (256, 512, 256 | 512):   # float epilogue / 64-filter workgroups / both, on top of YL_VARIANT_DEFAULT
        net.set_variant(common.VARIANT_DEFAULT | bits)
        b = net.predict(x).copy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), bits
    net.close()


@pytest.mark.parametrize("width,height,batch", [(64, 48, 2), (96, 96, 3)])
def test_xnor_conv_shortcut_fusion_is_bit_identical(width, height, batch):
    """conv(xnor, bit path) + [shortcut] folded into one kernel, as the reference GPU path does
    (src/additionally.c:326-339): the shortcut tensor and everything behind it equal the unfused run bit for bit."""
    cfg, wts = _mixed_xnor_files(width, height)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 0, device=0)
    fused = Network.load(cfg, wts, batch, 0, device=0, fuse=True)
    plain.predict(x)
    fused.predict(x)
    infos = plain.layers()
    sc = [i for i, li in enumerate(infos) if li["type"] == common.SHORTCUT]
    assert len(sc) == 1 and "xnor" in fused.layer_kernel(sc[0] - 1)
    assert not fused.layer_materialised(sc[0] - 1)          # the conv's own tensor is not written
    for i in range(sc[0], plain.n):
        if fused.layer_materialised(i):
            assert np.array_equal(plain.layer_output(i).view(np.uint32), fused.layer_output(i).view(np.uint32)), i
    plain.close(); fused.close()
