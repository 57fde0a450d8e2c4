"""Pin the oracle restatement (oracle/yolo2_oracle.c) bit-for-bit against the
reference's own CPU path (oracle/_ref/libyolo2ref.so = unmodified reference
sources, golden build flags) on whole networks, every layer.

The reference ships no golden vectors, so this -- together with
tests/golden fixtures generated from the same library -- is what makes the
oracle trustworthy.  Skipped only if oracle/_ref was never built.
"""
import numpy as np
import pytest

import common
from common import OracleNet, Network, refbind

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built (needs /root/reference once)")

CASES = [
    # name, width, height, batch, quantized
    ("yolov3-tiny", 96, 96, 2, 0),
    ("yolov3-tiny", 64, 96, 1, 1),
    ("yolov3", 64, 64, 2, 0),
    ("yolov3", 64, 64, 1, 1),
    ("tiny-yolo-xnor", 96, 96, 2, 0),
    ("tiny-yolo-xnor", 160, 128, 1, 0),
]


@pytest.mark.parametrize("name,width,height,batch,quantized", CASES)
def test_oracle_matches_reference_every_layer(olib, name, width, height, batch, quantized):
    cfg, wts = common.model_files(name, width, height)
    ref = refbind.RefNetwork(cfg, wts, batch, quantized)
    net = Network.load(cfg, wts, batch, quantized)
    x = common.seeded_input(batch, 3, height, width)
    ref.predict(x)
    on = OracleNet(net, olib)
    on.set_route_inputs(open(cfg).read())
    on.forward(x)
    assert ref.n == net.n
    for i in range(net.n):
        a = on.outputs[i]
        b = ref.layer_output(i)
        assert a.shape == b.shape
        # bit-exact: the restatement must be the same arithmetic, not merely close
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if not same:
            bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
            raise AssertionError("layer %d (%s): %d/%d values differ, first idx %d: oracle %r ref %r" % (
                i, net.layer_info(i), bad.size, a.size, bad[0], a[bad[0]], b[bad[0]]))


def test_maxpool_op_pin(olib):
    rng = np.random.default_rng(7)
    for size, stride, w, h in [(2, 2, 8, 6), (2, 1, 13, 13), (5, 1, 9, 7), (9, 1, 13, 13), (13, 1, 13, 13), (3, 2, 11, 9)]:
        pad = size - 1
        ow, oh = (w + pad - size) // stride + 1, (h + pad - size) // stride + 1
        c, b = 3, 2
        x = rng.standard_normal(b * c * h * w).astype(np.float32)
        a = np.zeros(b * c * oh * ow, dtype=np.float32)
        r = np.zeros_like(a)
        olib.oracle_maxpool(common.fp(x), common.fp(a), size, w, h, ow, oh, c, pad, stride, b)
        rl = refbind._bind(refbind.GOLD)
        rl.ref_maxpool(common.fp(x), common.fp(r), size, w, h, ow, oh, c, pad, stride, b)
        assert np.array_equal(a, r), (size, stride)


MORE_CASES = [
    ("all-activations", 48, 32, 2, 0),     # every activation of activate() in conv / shortcut / xnor layers
    ("all-activations", 48, 32, 1, 1),     # -quantized: the INT8 convolution undoes nothing but LEAKY
    ("yolov2-voc", 96, 96, 2, 0),          # reorg + region (softmax) + multi-input route
    ("yolov2-voc", 96, 96, 1, 1),
    ("tiny-yolo-voc", 96, 64, 2, 0),
    ("yolov3-spp", 64, 64, 1, 0),          # stride-1 max-pools 5/9/13
]


@pytest.mark.parametrize("name,width,height,batch,quantized", MORE_CASES)
def test_oracle_matches_reference_other_cfgs(olib, name, width, height, batch, quantized):
    test_oracle_matches_reference_every_layer(olib, name, width, height, batch, quantized)


@pytest.mark.parametrize("batch", [1, 2])
def test_oracle_matches_reference_xnor_fallback(olib, batch):
    """custom cfg mixing xnor bit-path and FP32-fallback convs (stride 2, 1x1) + shortcut"""
    import os
    from yolo2_light_amd import weights as W
    sys_path_cfg = os.path.join(common.workdir(), "xnor-mixed-pin.cfg")
    import test_gpu_int8_xnor as T
    text = T.XNOR_MIXED_CFG % (64, 48)
    open(sys_path_cfg, "w").write(text)
    wts = sys_path_cfg[:-4] + ".weights"
    W.write_synthetic_weights(text, wts, seed=3)
    common._MODEL_CACHE[("xnor-mixed-pin", 64, 48, 1)] = (sys_path_cfg, wts)
    test_oracle_matches_reference_every_layer(olib, "xnor-mixed-pin", 64, 48, batch, 0)


IMAGE_CASES = [
    # source w, h -> network w, h
    (768, 576, 416, 416),      # the shape of bin/dog.jpg into tiny-416: downscale, aspect change
    (640, 480, 608, 608),
    (333, 500, 608, 608),      # upscale one axis, downscale the other
    (100, 60, 416, 416),       # pure upscale
    (416, 416, 416, 416),      # same size (resize_image still runs, src/main.c:189)
    (1920, 1080, 608, 608),
    (1, 7, 32, 32),            # im.w == 1 edge branch
    (9, 1, 32, 32),            # im.h == 1 edge branch
    (37, 23, 64, 96),
]


@pytest.mark.parametrize("sw,sh,w,h", IMAGE_CASES)
def test_image_front_end_matches_reference(olib, sw, sh, w, h):
    """u8 HWC -> float CHW /255. -> resize_image, against the reference's own load_image +
    resize_image run on a PPM of the same pixels."""
    import ctypes as C
    import os
    rng = np.random.default_rng(sw * 1000 + sh)
    pix = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
    path = os.path.join(common.workdir(), "img_%dx%d.ppm" % (sw, sh))
    common.write_ppm(path, pix)
    ref = np.zeros((3, h, w), dtype=np.float32)
    rw, rh = C.c_int(0), C.c_int(0)
    rl = refbind._bind(refbind.GOLD)
    assert rl.ref_load_resized(path.encode(), w, h, common.fp(ref), C.byref(rw), C.byref(rh)) == 0
    assert (rw.value, rh.value) == (sw, sh)
    got = common.oracle_load_resized(olib, pix, w, h)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("seed,n,scale,shape", [
    (0, 200000, 1.0, "halfnormal"), (1, 50000, 6.0, "halfnormal"), (2, 300000, 0.3, "leaky"),
    (3, 20000, 30.0, "uniform"), (4, 150000, 2.0, "leaky"), (5, 1000, 1.0, "halfnormal"),
])
def test_entropy_calibration_matches_reference(olib, seed, n, scale, shape):
    """the calibration tool's multiplier search (entropy_calibration(x, n, 1/16, 4096)) bit for bit"""
    rng = np.random.default_rng(seed)
    if shape == "uniform":
        x = rng.uniform(0, scale, n)
    elif shape == "leaky":
        x = rng.standard_normal(n) * scale
        x = np.where(x > 0, x, 0.1 * x)
    else:
        x = np.abs(rng.standard_normal(n)) * scale
    x = x.astype(np.float32)
    rl = refbind._bind(refbind.GOLD)
    ref = rl.ref_entropy_calibration(common.fp(x), x.size, 1.0 / 16, 4096)
    got = olib.oracle_entropy_calibration(common.fp(x), x.size, 1.0 / 16, 4096)
    assert np.float32(got).view(np.uint32) == np.float32(ref).view(np.uint32), (got, ref)


def test_fast_int8_oracle_equals_oracle(olib):
    """oracle/fast_oracle.c (tap-outermost integer accumulation, OpenMP) == oracle_conv_int8 bit for bit:
    accumulators and outputs, incl. stride 2, 1x1, no padding, the int16 wrap corner and the clamp."""
    import ctypes as C
    fast = common.oracle_fast_lib()
    i8p, i32p = C.POINTER(C.c_int8), C.POINTER(C.c_int32)
    rng = np.random.default_rng(4)
    for (B, Cc, H, W, M, size, stride, pad) in [(2, 16, 13, 11, 24, 3, 1, 1), (1, 32, 19, 23, 20, 3, 2, 1),
                                                (2, 64, 9, 7, 33, 1, 1, 0), (1, 8, 10, 10, 6, 3, 1, 0),
                                                (1, 5, 12, 9, 7, 5, 2, 2), (1, 2048, 4, 4, 3, 3, 1, 1)]:
        K = Cc * size * size
        wts = rng.normal(0, 0.7 if Cc < 2048 else 4.0, M * K).astype(np.float32)
        wq = np.zeros(M * K, np.int8)
        w_mult = olib.oracle_quantize_weights(common.fp(wts), M * K, wq.ctypes.data_as(i8p))
        if Cc == 2048:
            wq[:] = np.where(rng.random(M * K) < 0.5, 127, 120).astype(np.int8)     # drives |acc/32| past 32767
        bias = rng.normal(0, 0.5, M).astype(np.float32)
        in_mult = 11.3
        x = (rng.standard_normal((B, Cc, H, W)) * (3 if Cc < 2048 else 40)).astype(np.float32)
        x.reshape(-1)[:6] = [1e9, -1e9, 40000.7 / in_mult, -33000.2 / in_mult, 127.9 / in_mult, -128.5 / in_mult]
        oh, ow = (H + 2 * pad - size) // stride + 1, (W + 2 * pad - size) // stride + 1
        a = np.zeros(B * M * oh * ow, np.float32); b = np.zeros_like(a)
        aa = np.zeros(a.size, np.int32); ba = np.zeros(a.size, np.int32)
        olib.oracle_conv_int8(common.fp(x), wq.ctypes.data_as(i8p), common.fp(bias), common.fp(a), aa.ctypes.data_as(i32p),
                              B, Cc, H, W, M, size, stride, pad, 7, in_mult, w_mult)
        fast.oracle_conv_int8_fast(common.fp(x), wq.ctypes.data_as(i8p), common.fp(bias), common.fp(b),
                                   ba.ctypes.data_as(i32p), B, Cc, H, W, M, size, stride, pad, 7, in_mult, w_mult)
        assert np.array_equal(aa, ba)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if Cc == 2048:
            assert (np.abs(aa) == 32767).any(), "clamp corner not reached"
