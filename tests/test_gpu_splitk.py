"""Split K (yl_network_set_split_k): FP32 convolutions whose grid leaves CUs idle -- yolov3's 19 x 19 and
38 x 38 layers at 8 images per GPU, the regime config 3 runs at 8 GPUs -- cut their input channels into 2-4 ranges; a second kernel
adds the partial sums in range order.  The reference has no analogue (one image, one device, src/main.c:653-661); what is pinned:
the result against the oracle under the FP32 contract, that the split really ran, and that it is deterministic (bit-identical
from run to run and from network to network).  It is NOT bit-equal to the unsplit layer (another summation order) -- which is why
it is opt-in and the batch-B == batch-1 tests of tests/test_gpu_headline.py run without it."""
import numpy as np
import pytest

import common
import descs as D
from common import Network, fp, fp32_close, refbind

pytestmark = pytest.mark.gpu

# (B, C, H, W, M, size, stride, act): the small-grid layers of yolov3-608 at 8 images per GPU (3 where the scalar oracle would take half a minute)
SHAPES = [
    (8, 1024, 19, 19, 512, 1, 1, D.LEAKY),      # K1x 1x1, 64 channel blocks: four ranges
    (8, 512, 19, 19, 256, 1, 1, D.LEAKY),       # K1x 1x1 below one workgroup per CU
    (3, 512, 19, 19, 1024, 3, 1, D.LEAKY),      # K1r, 32 channel blocks x 3 filter rows
    (8, 256, 19, 19, 512, 3, 1, D.LINEAR),      # K1r, 16 channel blocks
    (3, 512, 38, 38, 1024, 3, 2, D.LEAKY),      # K1x 3x3 / stride 2
    (3, 528, 13, 13, 70, 1, 1, D.LEAKY),        # 33 channel blocks: three ranges of 11; M not a multiple of the tile
    (2, 1024, 13, 13, 255, 1, 1, D.LINEAR),     # a linear head without its [yolo]
]


def _net(shape, wts, bias, split):
    B, Cc, H, W, M, size, stride, act = shape
    d = D.conv(B, W, H, Cc, M, size, stride, size // 2, act, wts, bias)
    net = Network.from_desc([d], B, W, H, Cc, 0)
    net.set_split_k(split)
    net.to_device(0)
    return net, d


@pytest.mark.parametrize("shape", SHAPES)
def test_split_k_layer_vs_oracle(olib, shape):
    B, Cc, H, W, M, size, stride, act = shape
    rng = np.random.default_rng(77 + Cc + M)
    K = Cc * size * size
    wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
    bias = rng.normal(0, 0.5, M).astype(np.float32)
    x = (rng.standard_normal((B, Cc, H, W)) * np.exp(rng.uniform(-4, 2, (B, Cc, H, W)))).astype(np.float32)
    a, d = _net(shape, wts, bias, True)
    b, _ = _net(shape, wts, bias, False)
    ya = a.predict(x).copy()
    yb = b.predict(x).copy()
    assert ",split" in a.layer_kernel(0) and ",split" not in b.layer_kernel(0), (a.layer_kernel(0), b.layer_kernel(0))
    ref = np.zeros(B * d.outputs, dtype=np.float32)
    olib.oracle_conv_f32(fp(x), fp(wts), fp(bias), fp(ref), B, Cc, H, W, M, size, stride, size // 2, act)
    for tag, y in (("split", ya), ("unsplit", yb)):
        ok, ratio, worst = fp32_close(y, ref)
        assert ok, "%s %r: err/allowed %.3g at %d: got %r ref %r" % (tag, shape, ratio, worst, y[worst], ref[worst])
        assert ratio < 0.25
    # deterministic: the same bits from a second pass and from a second network
    assert np.array_equal(a.predict(x).view(np.uint32), ya.view(np.uint32))
    c, _ = _net(shape, wts, bias, True)
    assert np.array_equal(c.predict(x).view(np.uint32), ya.view(np.uint32)) and c.layer_kernel(0) == a.layer_kernel(0)
    a.close(); b.close(); c.close()


def test_split_k_leaves_full_grids_alone():
    shape = (8, 512, 38, 38, 256, 1, 1, D.LEAKY)            # 38 x 38 at 8 images: 724 workgroups, 32 panels -- the second stage would cost more
    B, Cc, H, W, M, size, stride, act = shape
    rng = np.random.default_rng(5)
    wts = rng.normal(0, 0.03, M * Cc * size * size).astype(np.float32)
    bias = np.zeros(M, np.float32)
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    a, _ = _net(shape, wts, bias, True)
    b, _ = _net(shape, wts, bias, False)
    ya, yb = a.predict(x).copy(), b.predict(x).copy()
    assert ",split" not in a.layer_kernel(0) and a.layer_kernel(0) == b.layer_kernel(0)
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    a.close(); b.close()


def test_yolov3_608_batch8_split_k_against_the_reference_library():
    """config 3's per-GPU shard at 8 GPUs (8 images, fusion on, split K on) against network_predict_cpu of the unmodified reference:
    image 5, every materialised tensor at the FP32 contract's fp32_close; the split layers are named; two passes give the same bits."""
    common.require_ref()
    size, B, img = 608, 8, 5
    cfg, wts = common.model_files("yolov3", size, size)
    x = common.seeded_input(B, 3, size, size)
    net = Network.load(cfg, wts, B, 0, device=0, fuse=True, split_k=True)
    import os
    old = os.environ.get("YL_PREDICT_SPLIT")
    os.environ["YL_PREDICT_SPLIT"] = "1"                    # one pass over the 8 images (the grid the ranges are chosen for)
    try:
        head = net.predict(x).copy()
        again = net.predict(x).copy()
    finally:
        if old is None:
            os.environ.pop("YL_PREDICT_SPLIT", None)
        else:
            os.environ["YL_PREDICT_SPLIT"] = old
    assert np.array_equal(head.view(np.uint32), again.view(np.uint32))
    split_layers = [i for i in range(net.n) if ",split" in net.layer_kernel(i)]
    print("split layers:", [(i, net.layer_kernel(i)) for i in split_layers])
    assert len(split_layers) >= 15
    ref = refbind.RefNetwork(cfg, wts, 1, 0)
    ref.predict(x[img:img + 1])
    n = 0
    for i in range(net.n):
        if not net.layer_materialised(i):
            continue
        ok, ratio, worst = fp32_close(net.layer_output_image(i, img), ref.layer_output(i))
        assert ok, "layer %d %s: err/allowed %.3g" % (i, net.layer_kernel(i), ratio)
        n += 1
    assert n >= 55
    r = ref.get_detections(0, size, size, 0.24, nms=0.4)
    g = net.get_boxes(img, size, size, 0.24, nms=0.4, relative=1)
    assert abs(len(r) - len(g)) <= 2
    net.close()
