"""K1r (conv_f32_row3.hip): the 3x3 / stride-1 / pad-1 FP32 convolution as row-wise Winograd F(2,3) on the BF16 matrix pipe
with three-piece operands -- against the oracle restatement of forward_convolutional_layer_cpu
(src/yolov2_forward_network.c:204-261), against a float64 convolution, and for the properties the other FP32 kernels hold:
every tile / schedule bit-identical to every other, fused [shortcut] == the unfused pair bit for bit, batch items
independent of what shares their workgroup.
"""
import numpy as np
import pytest

import common
import descs as D
from common import Network, fp, fp32_close

pytestmark = pytest.mark.gpu

ROW3_TILES = list(range(61, 71))         # yl_network_set_conv_tile: 61..69 = conv_f32_row3.hip's tiles and schedules, 70 = its view form

ROW3_SHAPES = [
    # B, C, H, W, M, act
    (2, 16, 13, 13, 33, D.LEAKY),          # one channel block (3 groups), M tail, odd width: 7 tiles per row, the last half empty
    (3, 32, 19, 19, 70, D.LINEAR),         # odd size 19 (yolov3-608's last scale), linear
    (1, 256, 13, 13, 512, D.LEAKY),        # deep K: 48 groups, 4 filter tiles of 128
    (2, 48, 11, 9, 96, D.LEAKY),           # H != W, both odd, three channel blocks
    (2, 64, 38, 38, 128, D.LEAKY),         # even width, several tile blocks
    (1, 32, 76, 76, 64, D.LEAKY),          # many tile blocks, M = 64
    (4, 128, 6, 10, 255, D.LINEAR),        # M = 255, a tile block spans images
    (9, 16, 5, 5, 16, D.LEAKY),            # 15 tiles per image: a block of 128 tiles spans 9 images, mostly empty
    (1, 16, 4, 4, 8, D.LEAKY),             # the smallest layer the dispatcher sends here
    (2, 32, 5, 125, 40, D.LEAKY),          # 63 tiles per row: the widest map the view form takes (128 + 2 * 63 = 254 entries), odd width
]


def _layer(shape, seed):
    B, Cc, H, W, M, act = shape
    rng = np.random.default_rng(seed + M + H)
    K = Cc * 9
    wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
    bias = rng.normal(0, 0.5, M).astype(np.float32)
    # nine octaves of magnitude: every piece of the split carries signal
    x = (rng.standard_normal((B, Cc, H, W)) * np.exp(rng.uniform(-6, 3, (B, Cc, H, W)))).astype(np.float32)
    return wts, bias, x


def _single(shape, wts, bias, variant=0):
    B, Cc, H, W, M, act = shape
    d = D.conv(B, W, H, Cc, M, 3, 1, 1, act, wts, bias)
    net = Network.from_desc([d], B, W, H, Cc, 0)
    net.set_variant(variant)
    net.to_device(0)
    return net, d


@pytest.mark.parametrize("shape", ROW3_SHAPES)
@pytest.mark.parametrize("tile", [61, 62, 63, 64, 65, 66, 67, 68, 70])
def test_conv_row3_vs_oracle(olib, shape, tile):
    B, Cc, H, W, M, act = shape
    wts, bias, x = _layer(shape, 2718)
    net, d = _single(shape, wts, bias)
    net.set_conv_tile(tile)
    got = net.predict(x).copy()
    assert "conv_f32_row3<" in net.layer_kernel(0), net.layer_kernel(0)
    ref = np.zeros(B * d.outputs, dtype=np.float32)
    olib.oracle_conv_f32(fp(x), fp(wts), fp(bias), fp(ref), B, Cc, H, W, M, 3, 1, 1, act)
    ok, ratio, worst = fp32_close(got, ref)
    assert ok, "tile %d shape %r: err/allowed %.3g at %d: got %r ref %r" % (tile, shape, ratio, worst, got[worst], ref[worst])
    assert ratio < 0.2          # FP32-roundoff class, like the FP32-MFMA and the 2-D Winograd kernel
    # against a float64 convolution: not farther from the truth than 1.5x the FP32-MFMA kernel on the same layer
    net.set_conv_tile(14)
    direct = net.predict(x).copy()
    assert "row3" not in net.layer_kernel(0)
    import torch
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wts.reshape(M, Cc, 3, 3)).double(),
                                       torch.from_numpy(bias).double(), stride=1, padding=1)
    if act == D.LEAKY:
        truth = torch.where(truth > 0, truth, 0.1 * truth)
    truth = truth.numpy().reshape(-1)
    rms = float(np.sqrt(np.mean(truth ** 2)))
    e_r3 = float(np.sqrt(np.mean((got.astype(np.float64) - truth) ** 2))) / rms
    e_f32 = float(np.sqrt(np.mean((direct.astype(np.float64) - truth) ** 2))) / rms
    assert e_r3 <= 1.5 * e_f32 + 1e-9, "shape %r: rms error vs float64 %.3g (row3) vs %.3g (FP32 MFMA)" % (shape, e_r3, e_f32)
    net.close()


@pytest.mark.parametrize("shape", ROW3_SHAPES)
def test_row3_tiles_and_schedules_bit_identical(shape):
    """ten instances (workgroup tile, planes per panel, where the barrier sits, V staged per filter row or once per channel block): the same
    products in the same order"""
    wts, bias, x = _layer(shape, 31415)
    net, _ = _single(shape, wts, bias)
    base = None
    names = set()
    for tile in ROW3_TILES:
        net.set_conv_tile(tile)
        got = net.predict(x).copy()
        names.add(net.layer_kernel(0))
        if base is None:
            base = got
        assert np.array_equal(got.view(np.uint32), base.view(np.uint32)), "tile %d shape %r" % (tile, shape)
    assert len(names) == len(ROW3_TILES) and all("conv_f32_row3<" in n for n in names), names
    net.close()


@pytest.mark.parametrize("width,height,act", [(38, 38, D.LEAKY), (19, 19, D.LEAKY), (13, 9, D.LINEAR)])
def test_row3_fused_shortcut_is_bit_identical(width, height, act):
    """conv(1x1) -> conv(3x3, K1r) -> [shortcut]: the epilogue that adds the residual operand (out_add only) against the
    two-kernel form, and the heuristic picks K1r for the 3x3 layer with the default variant"""
    B, Cc, M = 3, 64, 64
    rng = np.random.default_rng(5)
    w1 = rng.normal(0, np.sqrt(2.0 / Cc), Cc * Cc).astype(np.float32)
    w2 = rng.normal(0, np.sqrt(2.0 / (Cc * 9)), M * Cc * 9).astype(np.float32)
    b1 = rng.normal(0, 0.5, Cc).astype(np.float32)
    b2 = rng.normal(0, 0.5, M).astype(np.float32)
    x = rng.standard_normal((B, Cc, height, width)).astype(np.float32)

    def build(fuse):
        descs = [D.conv(B, width, height, Cc, Cc, 1, 1, 0, D.LEAKY, w1, b1),
                 D.conv(B, width, height, Cc, M, 3, 1, 1, act, w2, b2),
                 D.shortcut(B, 0, (width, height, Cc), (width, height, M))]
        net = Network.from_desc(descs, B, width, height, Cc, 0)
        net.set_fusion(fuse)
        net.to_device(0)
        return net

    plain, fused = build(False), build(True)
    a, b = plain.predict(x).copy(), fused.predict(x).copy()
    assert "conv_f32_row3<" in plain.layer_kernel(1) and "conv_f32_row3<" in fused.layer_kernel(1)
    assert not fused.layer_materialised(1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    plain.close(); fused.close()


def test_row3_view_form_falls_back_on_wide_maps():
    """more than 63 tiles per row: tile 70 runs the pinned 128 x 128 tile (a view would span more than 254 entries), same bits"""
    shape = (1, 16, 4, 130, 24, D.LEAKY)
    wts, bias, x = _layer(shape, 99)
    net, _ = _single(shape, wts, bias)
    net.set_conv_tile(61)
    a = net.predict(x).copy()
    net.set_conv_tile(70)
    b = net.predict(x).copy()
    assert net.layer_kernel(0) == "conv_f32_row3<128x128t,pipe>", net.layer_kernel(0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    net.close()


def test_row3_batch_items_independent():
    """image k of a batch of 5 == the same image alone, bit for bit (tiles of several images share workgroups at 13 x 13)"""
    shape = (5, 32, 13, 13, 96, D.LEAKY)
    wts, bias, x = _layer(shape, 1618)
    net, _ = _single(shape, wts, bias, variant=-1)
    got = net.predict(x).copy().reshape(5, -1)
    assert "conv_f32_row3<" in net.layer_kernel(0)
    net.close()
    one = (1,) + shape[1:]
    net1, _ = _single(one, wts, bias, variant=-1)
    for k in (0, 2, 4):
        alone = net1.predict(x[k:k + 1]).copy().reshape(-1)
        assert np.array_equal(alone.view(np.uint32), got[k].view(np.uint32)), "image %d" % k
    net1.close()


def test_row3_whole_network_yolov3():
    """yolov3 with the default variant (K1r on the 3x3 / stride-1 layers, K1x on the direct ones) against the same network
    with the 2-D FP32 Winograd kernel: every materialised tensor within the FP32 contract, the same boxes, and fused ==
    unfused bit for bit"""
    name, width, height, batch = "yolov3", 160, 96, 2
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    ref = Network.load(cfg, wts, batch, 0, device=0, fuse=True, variant=62 | 1024)
    a = Network.load(cfg, wts, batch, 0, device=0, fuse=True)
    b = Network.load(cfg, wts, batch, 0, device=0, fuse=False)
    ref.predict(x); a.predict(x); b.predict(x)
    kernels = [a.layer_kernel(i) for i in range(a.n)]
    assert sum("conv_f32_row3<" in k for k in kernels) >= 20, kernels       # (the 5 x 3 maps of the last scale are below the kernel's 4 x 4)
    assert not any("conv_f32_row3<" in ref.layer_kernel(i) for i in range(ref.n))
    for i in range(a.n):
        if not a.layer_materialised(i):
            continue
        ya, yb = a.layer_output(i), b.layer_output(i)
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), "fused vs unfused, layer %d" % i
        if ref.layer_materialised(i):
            ok, ratio, worst = fp32_close(ya, ref.layer_output(i))
            assert ok, "layer %d (%s): err/allowed %.3g" % (i, kernels[i], ratio)
    for im in range(batch):
        ra, rr = a.get_boxes(im, width, height, 0.24, nms=0.4), ref.get_boxes(im, width, height, 0.24, nms=0.4)
        assert ra.shape == rr.shape and np.allclose(ra, rr, rtol=1e-4, atol=1e-5)
        assert np.array_equal(ra, b.get_boxes(im, width, height, 0.24, nms=0.4))
    # [upsample] -> [route] -> conv 1x1 (layers 85-87, 97-99): under fusion K1x reads the two sources directly, the upsampled tensor
    # and the concatenation are not written (and everything behind them equalled the unfused run above)
    two = [i for i in range(a.n) if "up+route" in kernels[i]]
    assert len(two) == 2 and not any("up+route" in b.layer_kernel(i) for i in range(b.n)), kernels
    for i in two:
        assert not a.layer_materialised(i - 1) and not a.layer_materialised(i - 2) and b.layer_materialised(i - 1) and b.layer_materialised(i - 2)
    # ... and a knob that moves those convolutions to another kernel brings the two layers back in the same pass (same bits)
    head = a.predict(x).copy()
    a.set_conv_tile(14)
    again = a.predict(x).copy()
    assert all(a.layer_materialised(i - 1) and a.layer_materialised(i - 2) and "x3" not in a.layer_kernel(i) for i in two)
    li, ls = a.layer_info(two[0] - 1), a.layer_info(two[0] - 3)             # the [route] and the convolution in front of the [upsample]
    cat = a.layer_output(two[0] - 1).reshape(batch, li["out_c"], li["out_h"], li["out_w"])
    src = a.layer_output(two[0] - 3).reshape(batch, ls["out_c"], ls["out_h"], ls["out_w"])
    assert np.array_equal(cat[:, :ls["out_c"]].view(np.uint32), src.repeat(2, axis=2).repeat(2, axis=3).view(np.uint32))
    a.set_conv_tile(0)
    assert np.array_equal(head.view(np.uint32), a.predict(x).view(np.uint32))
    ok, ratio, _ = fp32_close(again, head)
    assert ok, ratio
    ref.close(); a.close(); b.close()


def test_row3_view_form_by_variant_bit_is_bit_identical():
    """variant bit 13: the heuristic takes the view form wherever it would take the pinned 128 x 128 tile (a grid that fills the chip
    several times over), and nowhere else; same bits as the default variant"""
    shape = (8, 64, 76, 76, 512, D.LEAKY)
    wts, bias, x = _layer(shape, 4)
    a, _ = _single(shape, wts, bias, variant=-1)
    v, _ = _single(shape, wts, bias, variant=common.VARIANT_DEFAULT | 8192)
    ya, yv = a.predict(x).copy(), v.predict(x).copy()
    assert a.layer_kernel(0) == "conv_f32_row3<128x128t,pipe>" and v.layer_kernel(0) == "conv_f32_row3<128x128t,view>", (a.layer_kernel(0), v.layer_kernel(0))
    assert np.array_equal(ya.view(np.uint32), yv.view(np.uint32))
    a.close(); v.close()
    small = (2, 32, 19, 19, 70, D.LINEAR)                # below the chip: the 64 x 64 tile either way
    wts, bias, x = _layer(small, 5)
    v, _ = _single(small, wts, bias, variant=common.VARIANT_DEFAULT | 8192)
    v.predict(x)
    assert "view" not in v.layer_kernel(0) and "conv_f32_row3<" in v.layer_kernel(0), v.layer_kernel(0)
    v.close()


@pytest.mark.parametrize("size,tile", [(1, 51), (3, 61)])
def test_three_piece_kernels_at_the_edges_of_the_format(olib, size, tile):
    """The three-piece split (K1x, K1r) at the edges of FP32 (VERDICT round 4, weak 1 iii): up to 3.38e38 -- the largest
    magnitude whose first bf16 piece is finite -- an input behaves like any other and the result matches the oracle; in the top
    sliver above it (|x| >= 3.3961e38, 0.2 % of the top binade) the first piece rounds to Inf and the output is not finite where the reference's FP32
    product still is: a documented deviation, irrelevant for activations.  Sub-normal inputs lose their
    residual pieces (flushed): the result stays within an absolute 1e-37 of the oracle's."""
    B, Cc, H, W, M = 1, 16, 8, 8, 32
    rng = np.random.default_rng(12)
    K = Cc * size * size
    wts = rng.normal(0, 0.05, M * K).astype(np.float32)
    bias = np.zeros(M, np.float32)
    d = D.conv(B, W, H, Cc, M, size, 1, size // 2, D.LINEAR, wts, bias)
    net = Network.from_desc([d], B, W, H, Cc, 0)
    net.set_variant(0)
    net.to_device(0)
    net.set_conv_tile(tile)

    def run(x):
        got = net.predict(x).copy()
        assert ("conv_f32_x3<" if size == 1 else "conv_f32_row3<") in net.layer_kernel(0)
        ref = np.zeros(B * d.outputs, dtype=np.float32)
        olib.oracle_conv_f32(fp(x), fp(wts), fp(bias), fp(ref), B, Cc, H, W, M, size, 1, size // 2, D.LINEAR)
        return got, ref

    # (a) one huge but splittable value per image plane position: finite, and as close as any other input
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    x[0, 3, 4, 4] = np.float32(3.38e38)                 # times |w| ~ 0.05 stays below FP32's maximum
    x[0, 5, 2, 6] = np.float32(-2.0e36)
    got, ref = run(x)
    assert np.all(np.isfinite(got)) and np.all(np.isfinite(ref))
    ok, ratio, worst = fp32_close(got, ref)
    assert ok, "err/allowed %.3g" % ratio
    # (b) the top half-binade: the reference stays finite, the split's first piece is Inf
    x2 = x.copy()
    x2[0, 3, 4, 4] = np.float32(3.4e38)                 # past the rounding boundary 3.3961e38 of bf16's largest finite value
    got2, ref2 = run(x2)
    assert np.all(np.isfinite(ref2)), "0.05 * 3.4e38 is a finite FP32 product"
    assert not np.all(np.isfinite(got2)), "documented deviation: bf16(3.4e38) is Inf"
    assert np.isfinite(got2).sum() >= got2.size // 2, "only outputs that see the element are affected"
    # (c) sub-normal inputs
    x3 = (rng.standard_normal((B, Cc, H, W)) * 1e-40).astype(np.float32)
    got3, ref3 = run(x3)
    assert np.all(np.isfinite(got3)) and float(np.max(np.abs(got3.astype(np.float64) - ref3))) < 1e-37
    net.close()


def test_two_source_plan_with_pool_fusion_around_it():
    """The planner orders: the two-source plan ([upsample] -> [route] -> conv 1x1 read from the two tensors) is made
    BEFORE the [maxpool]-fusion plan.  A network with a pool-fused convolution as the route's second source and a [maxpool] BEHIND the
    1x1 convolution: the two-source form is taken, the pooled / full tensors its neighbours need exist, and every materialised tensor
    equals the unfused run bit for bit."""
    B, W, H = 2, 16, 16
    rng = np.random.default_rng(41)

    def cw(c, m, k):
        return rng.normal(0, np.sqrt(2.0 / (c * k * k)), m * c * k * k).astype(np.float32), rng.normal(0, 0.3, m).astype(np.float32)
    w0, b0 = cw(16, 32, 3); w2, b2 = cw(32, 32, 3); w5, b5 = cw(64, 96, 1)
    descs = [
        D.conv(B, W, H, 16, 32, 3, 1, 1, D.LEAKY, w0, b0),               # 0: also the route's second source -> its full tensor stays
        D.maxpool(B, W, H, 32, 2, 2),                                    # 1: foldable into layer 0 (K1w pooled form)
        D.conv(B, W // 2, H // 2, 32, 32, 3, 1, 1, D.LEAKY, w2, b2),     # 2
        D.upsample(B, W // 2, H // 2, 32, 2),                            # 3
        D.route(B, [3, 0], [32 * W * H, 32 * W * H], (W, H, 64)),        # 4
        D.conv(B, W, H, 64, 96, 1, 1, 0, D.LEAKY, w5, b5),               # 5: two-source candidate
        D.maxpool(B, W, H, 96, 2, 2),                                    # 6: behind a 1x1 convolution: never folded
    ]
    x = rng.standard_normal((B, 16, H, W)).astype(np.float32)
    nets = []
    for fuse in (True, False):
        net = Network.from_desc(descs, B, W, H, 16, 0)
        net.set_fusion(fuse)
        net.to_device(0)
        net.predict(x)
        nets.append(net)
    a, b = nets
    assert "up+route" in a.layer_kernel(5), a.layer_kernel(5)
    assert not a.layer_materialised(3) and not a.layer_materialised(4) and b.layer_materialised(3) and b.layer_materialised(4)
    assert a.layer_materialised(0) and a.layer_materialised(1) and a.layer_materialised(6)
    for i in range(a.n):
        if a.layer_materialised(i):
            assert np.array_equal(a.layer_output(i).view(np.uint32), b.layer_output(i).view(np.uint32)), "layer %d" % i
    a.close(); b.close()
