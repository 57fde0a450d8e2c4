"""the image front end on the GPU (yl_network_set_input_u8, csrc/preprocess.hip)
against the oracle restatement of load_image_stb's conversion + resize_image
(oracle_load_resized_u8, pinned bit-for-bit against the reference's own load_image/resize_image in
tests/test_oracle_pin.py::test_image_front_end_matches_reference).  Bar: bit-exact."""
import numpy as np
import pytest

import common
from common import Network

pytestmark = pytest.mark.gpu

SIZES = [(768, 576), (640, 480), (333, 500), (100, 60), (1920, 1080), (1, 7), (9, 1), (37, 23)]


@pytest.mark.parametrize("netw,neth", [(416, 416), (608, 608), (96, 64)])
def test_u8_front_end_bit_exact(olib, netw, neth):
    cfg, wts = common.model_files("yolov3-tiny", netw, neth)
    B = len(SIZES) + 1
    net = Network.load(cfg, wts, B, 0, device=0)
    rng = np.random.default_rng(netw)
    imgs = [rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8) for sw, sh in SIZES]
    imgs.append(rng.integers(0, 256, size=(neth, netw, 3), dtype=np.uint8))       # same size as the network
    for b, pix in enumerate(imgs):
        net.set_input_u8(b, pix)
    got = net.input_download()
    for b, pix in enumerate(imgs):
        ref = common.oracle_load_resized(olib, pix, netw, neth)
        assert np.array_equal(got[b].view(np.uint32), ref.view(np.uint32)), (b, pix.shape)
    net.close()


def test_frames_through_the_whole_path(olib):
    """u8 frames -> GPU front end -> forward -> batched detections  ==  oracle-resized float
    images -> predict -> host decode, bit for bit; slots are re-staged several times to exercise
    the staging ring (growth included)."""
    netw = neth = 416
    B = 4
    cfg, wts = common.model_files("yolov3-tiny", netw, neth)
    net = Network.load(cfg, wts, B, 0, device=0)
    rng = np.random.default_rng(3)
    for rnd, (sw, sh) in enumerate([(320, 240), (768, 576), (1280, 720)]):
        frames = [rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8) for _ in range(B)]
        for b, pix in enumerate(frames):
            net.set_input_u8(b, pix)
        net.forward_staged()
        rows, counts = net.get_boxes_batch(0.1, 0.45, cap=2048, sizes=(sw, sh), relative=0)
        heads_staged = [net.layer_output(i) for i in range(net.n) if net.layer_info(i)["type"] == common.YOLO]
        x = np.stack([common.oracle_load_resized(olib, pix, netw, neth) for pix in frames])
        net.predict(x)
        heads_ref = [net.layer_output(i) for i in range(net.n) if net.layer_info(i)["type"] == common.YOLO]
        for a, b_ in zip(heads_staged, heads_ref):
            assert np.array_equal(a.view(np.uint32), b_.view(np.uint32)), rnd
        for b in range(B):
            host = common.oracle_boxes(net, b, sw, sh, 0.1, nms=0.45, relative=0)
            assert counts[b] == len(host)
            assert np.array_equal(rows[b].view(np.uint32), host.view(np.uint32)), (rnd, b)
    net.close()


def test_batched_frames_equal_per_frame_calls():
    """yl_network_set_input_u8_batch (one call: pool-copied frames, uploads on the copy stream, resize kernels behind them) leaves
    the bits of one yl_network_set_input_u8 per frame in the input buffer -- mixed frame sizes, a partial range, the staging ring
    growing between rounds, batch calls and per-frame calls interleaved on the same slots."""
    netw, neth, B = 160, 96, 6
    cfg, wts = common.model_files("yolov3-tiny", netw, neth)
    net = Network.load(cfg, wts, B, 0, device=0)
    rng = np.random.default_rng(9)
    sizes = [(64, 48), (320, 240), (97, 33), (768, 576), (50, 70), (1280, 720)]
    for rnd in range(3):
        frames = [rng.integers(0, 256, size=(sizes[(b + rnd) % 6][1], sizes[(b + rnd) % 6][0], 3), dtype=np.uint8) for b in range(B)]
        for b, pix in enumerate(frames):
            net.set_input_u8(b, pix)
        want = net.input_download().copy()
        net.set_input_u8_batch([np.zeros_like(f) for f in frames])          # overwrite every slot, then the real frames in two ranges
        net.set_input_u8_batch(frames[:2])
        net.set_input_u8_batch(frames[2:], first=2)
        got = net.input_download()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), rnd
        net.set_input_u8(3, frames[0])                                       # a per-frame call behind a batch call on the same slot
        one = net.input_download()
        assert np.array_equal(one[3].view(np.uint32), want[0].view(np.uint32)) if frames[0].shape == frames[3].shape else True
    with pytest.raises(Exception):
        net.set_input_u8_batch(frames, first=1)                              # range beyond the batch
    net.close()


def test_front_end_rejects_bad_arguments():
    cfg, wts = common.model_files("yolov3-tiny", 96, 96)
    net = Network.load(cfg, wts, 2, 0, device=0)
    ok = np.zeros((10, 12, 3), dtype=np.uint8)
    with pytest.raises(Exception):
        net.set_input_u8(2, ok)                                  # slot out of range
    with pytest.raises(Exception):
        net.set_input_u8(0, np.zeros((10, 12, 1), dtype=np.uint8))   # channels != net.c
    net.set_input_u8(1, ok)
    net.close()
