"""the INT8 calibration tool (`darknet detector calibrate` -> network_calibrate_cpu ->
entropy_calibration).  CPU part: the library's KL scan from an exact histogram must give the
multiplier of the oracle restatement (itself pinned bit for bit against the reference's
entropy_calibration in tests/test_oracle_pin.py) on the same data."""
import ctypes as C

import numpy as np
import pytest

import common
from yolo2_light_amd._lib import lib

BIN_W, MAX_BIN = 1.0 / 16, 4096


def reference_histogram(x: np.ndarray) -> np.ndarray:
    """lround(fabs(x) / bin_width), saturated at max_bin - 1 (quantized.c:1306-1313), as counts"""
    v = np.abs(x.astype(np.float64)) / np.float64(np.float32(BIN_W))
    b = np.floor(v + 0.5).astype(np.int64)          # == lround for v >= 0
    b = np.minimum(b, MAX_BIN - 1)
    return np.bincount(b, minlength=MAX_BIN).astype(np.uint32)


@pytest.mark.parametrize("seed,n,scale,shape", [
    (0, 200000, 1.0, "halfnormal"), (1, 50000, 6.0, "halfnormal"), (2, 300000, 0.3, "leaky"),
    (3, 20000, 30.0, "uniform"), (4, 150000, 2.0, "leaky"), (6, 40000, 300.0, "halfnormal"),   # saturating tail
])
def test_kl_scan_from_histogram_equals_oracle(olib, seed, n, scale, shape):
    rng = np.random.default_rng(seed)
    if shape == "uniform":
        x = rng.uniform(0, scale, n)
    elif shape == "leaky":
        x = rng.standard_normal(n) * scale
        x = np.where(x > 0, x, 0.1 * x)
    else:
        x = np.abs(rng.standard_normal(n)) * scale
    x = x.astype(np.float32)
    want = olib.oracle_entropy_calibration(common.fp(x), x.size, BIN_W, MAX_BIN)
    h = reference_histogram(x)
    got = lib.yl_entropy_from_histogram(h.ctypes.data_as(C.POINTER(C.c_uint32)), MAX_BIN, BIN_W)
    assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), (got, want)


def test_kl_scan_rejects_bad_arguments():
    h = np.zeros(MAX_BIN, dtype=np.uint32)
    assert lib.yl_entropy_from_histogram(None, MAX_BIN, BIN_W) < 0
    assert lib.yl_entropy_from_histogram(h.ctypes.data_as(C.POINTER(C.c_uint32)), 64, BIN_W) < 0


# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,width,height,batch,n_images", [
    ("tiny-yolo-voc", 96, 96, 2, 4),          # conv/maxpool/region: the layer set the reference's tool knows
    ("yolov2-voc", 64, 64, 1, 3),             # + route, reorg; consecutive conv layers (the slot-0 quirk)
    ("yolov3-tiny", 96, 64, 2, 2),            # v3 layer types: real forward pass instead of skipping
])
def test_calibrate_equals_reference_procedure_on_gpu_activations(olib, name, width, height, batch, n_images):
    """yl_network_calibrate == entropy_calibration (oracle, pinned) applied to every conv layer's
    input as the GPU produced it, averaged with network_calibrate_cpu's slot arithmetic."""
    from common import Network
    cfg, wts = common.model_files(name, width, height)
    net = Network.load(cfg, wts, batch, 0, device=0)
    rng = np.random.default_rng(17)
    imgs = rng.random((n_images, 3, height, width), dtype=np.float32)
    got = net.calibrate(imgs)

    infos = net.layers()
    conv_ids = [i for i, li in enumerate(infos) if li["type"] == common.CONV]
    assert len(got) == len(conv_ids)
    per = {i: [] for i in conv_ids}            # per[layer] = multiplier of every image, in order
    for i0 in range(0, n_images, batch):
        x = imgs[i0:i0 + batch]
        net.predict(x)
        for i in conv_ids:
            src = x.reshape(batch, -1) if i == 0 else net.layer_output(i - 1).reshape(batch, -1)
            assert src.shape[1] == infos[i]["inputs"]
            for b in range(batch):
                row = np.ascontiguousarray(src[b])
                per[i].append(olib.oracle_entropy_calibration(common.fp(row), row.size, BIN_W, MAX_BIN))
    for k, i in enumerate(conv_ids):
        res = np.float32(0)
        prev_conv = i > 0 and infos[i - 1]["type"] == common.CONV
        res = np.float32(res + (np.float32(per[i - 1][n_images - 1]) if prev_conv else np.float32(0)))
        for j in range(1, n_images):
            res = np.float32(res + np.float32(per[i][j - 1]))
        want = np.float32(res / np.float32(n_images))
        assert np.float32(got[k]).view(np.uint32) == want.view(np.uint32), (name, i, got[k], want)
    net.close()


@pytest.mark.gpu
def test_calibrate_needs_fp32_network_and_whole_batches():
    from common import Network
    cfg, wts = common.model_files("yolov3-tiny", 64, 64)
    q = Network.load(cfg, wts, 1, 1, device=0)
    with pytest.raises(Exception):
        q.calibrate(np.zeros((1, 3, 64, 64), np.float32))
    q.close()
    f = Network.load(cfg, wts, 2, 0, device=0)
    with pytest.raises(Exception):
        f.calibrate(np.zeros((3, 3, 64, 64), np.float32))
    f.close()
