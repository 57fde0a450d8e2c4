"""Opt-in BF16 variant of the FP32 convolution (conv_bf16_mfma.hip, yl_network_set_precision).

Not inside the FP32 path's 1e-4 contract -- so it is pinned differently: products of two bf16 numbers are exact
in FP32, hence against the FP32 oracle applied to the ROUNDED operands the kernel may differ only by summation
order (tight bar, fp32_close); the distance to the un-rounded FP32 result is measured and bounded separately.
"""
import numpy as np
import pytest

import common
import descs as D
from common import Network, fp, fp32_close

pytestmark = pytest.mark.gpu


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bf16 -> float32 (what v_cvt_pk_bf16_f32 / the host weight packer do)"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


BF16_SHAPES = [
    # B, C, H, W, M, size, stride, pad, act
    (2, 16, 13, 13, 33, 3, 1, 1, D.LEAKY),         # M tail, odd size, tile spans images
    (1, 32, 26, 26, 64, 3, 2, 1, D.LEAKY),         # stride 2
    (3, 64, 7, 9, 255, 1, 1, 0, D.LINEAR),         # 1x1 head, M = 255
    (2, 128, 13, 13, 128, 1, 1, 0, D.LEAKY),       # 1x1, whole tap panels
    (1, 8, 12, 12, 24, 5, 1, 2, D.LEAKY),          # 5x5, one channel group
    (1, 256, 13, 13, 512, 3, 1, 1, D.LEAKY),       # deep K
    (2, 24, 11, 9, 96, 3, 1, 1, D.LEAKY),          # 3 groups -> padded to 4 (zero units)
    (5, 40, 5, 5, 70, 3, 1, 1, D.LINEAR),          # many tiny images in one N tile
]


@pytest.mark.parametrize("shape", BF16_SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_conv_bf16_vs_oracle_on_rounded_operands(olib, shape, tile):
    B, Cc, H, W, M, size, stride, pad, act = shape
    rng = np.random.default_rng(4321 + M + size)
    K = Cc * size * size
    wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
    bias = rng.normal(0, 0.5, M).astype(np.float32)
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    d = D.conv(B, W, H, Cc, M, size, stride, pad, act, wts, bias)
    net = Network.from_desc([d], B, W, H, Cc)
    net.set_precision(1)
    net.to_device(0)
    net.set_int8_tile(tile)
    got = net.predict(x)
    assert "bf16" in net.layer_kernel(0), net.layer_kernel(0)
    xr, wr = bf16_round(x), bf16_round(wts)
    ref = np.zeros(B * d.outputs, dtype=np.float32)
    olib.oracle_conv_f32(fp(xr), fp(wr), fp(bias), fp(ref), B, Cc, H, W, M, size, stride, pad, act)
    ok, ratio, worst = fp32_close(got, ref)
    assert ok and ratio < 0.2, "tile %d shape %r: err/allowed %.3g at %d: %r vs %r" % (tile, shape, ratio, worst, got[worst], ref[worst])
    # distance to the FP32 layer (un-rounded operands): operand rounding, ~2^-9 relative per product
    full = np.zeros_like(ref)
    olib.oracle_conv_f32(fp(x), fp(wts), fp(bias), fp(full), B, Cc, H, W, M, size, stride, pad, act)
    rel_rms = float(np.sqrt(np.mean((got - full) ** 2)) / (np.sqrt(np.mean(full ** 2)) + 1e-30))
    assert rel_rms < 1e-2, rel_rms
    net.close()


@pytest.mark.parametrize("name,width,height,batch", [("yolov3", 96, 96, 2), ("yolov3", 160, 96, 3), ("yolov3-tiny", 96, 96, 2)])
def test_bf16_network_teacher_forced_and_fusion(olib, name, width, height, batch):
    """Whole networks in BF16 mode.  Unfused: every BF16 conv is checked against the FP32 oracle conv applied to the
    bf16-rounded GPU input of that layer and rounded weights (summation order only).  Fused (conv+[shortcut], bf16
    side outputs, skipped FP32 tensors): every materialised tensor equals the unfused run bit for bit."""
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    plain = Network.load(cfg, wts, batch, 0, device=0, bf16=True)
    fused = Network.load(cfg, wts, batch, 0, device=0, fuse=True, bf16=True)
    plain.predict(x)
    fused.predict(x)
    infos = plain.layers()
    n_bf16 = 0
    for i, li in enumerate(infos):
        if li["type"] != common.CONV or "bf16" not in plain.layer_kernel(i):
            continue
        n_bf16 += 1
        xin = x if i == 0 else plain.layer_output(i - 1)
        w = plain.layer_weights(i)
        b = plain.layer_biases(i)
        ref = np.zeros(batch * li["outputs"], dtype=np.float32)
        olib.oracle_conv_f32(fp(bf16_round(xin)), fp(bf16_round(w)), fp(b), fp(ref), batch, li["c"], li["h"], li["w"],
                             li["n"], li["size"], li["stride"], li["pad"], li["activation"])
        ok, ratio, worst = fp32_close(plain.layer_output(i), ref)
        assert ok and ratio < 0.2, "layer %d: err/allowed %.3g" % (i, ratio)
    assert n_bf16 >= (10 if name == "yolov3-tiny" else 70)
    checked = 0
    for i, li in enumerate(infos):
        if fused.layer_materialised(i):
            assert np.array_equal(plain.layer_output(i).view(np.uint32), fused.layer_output(i).view(np.uint32)), "layer %d" % i
            checked += 1
    assert checked > 10
    for b in range(batch):
        assert np.array_equal(plain.get_boxes(b, width, height, 0.24, nms=0.4), fused.get_boxes(b, width, height, 0.24, nms=0.4))
    plain.close(); fused.close()


def test_bf16_vs_fp32_deviation_is_reported_and_bounded():
    """The opt-in's price on a whole network: head tensors of a BF16 run vs the FP32 run of the same images
    (synthetic weights).  Not a parity claim -- a bound that catches a broken BF16 path and documents the distance."""
    name, width, height, batch = "yolov3", 160, 160, 2
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    f32 = Network.load(cfg, wts, batch, 0, device=0, fuse=True)
    b16 = Network.load(cfg, wts, batch, 0, device=0, fuse=True, bf16=True)
    f32.predict(x)
    b16.predict(x)
    worst = 0.0
    for i, li in enumerate(f32.layers()):
        if li["type"] == common.YOLO:
            a, b = f32.layer_output(i), b16.layer_output(i)                # the head tensors the decode reads
            rel = float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(a ** 2)) + 1e-30))
            corr = float(np.corrcoef(a, b)[0, 1])
            print("head %d: rel rms %.3e corr %.6f" % (i, rel, corr))
            worst = max(worst, rel)
            assert corr > 0.999
    assert worst < 5e-2
    f32.close(); b16.close()
