"""batched get_network_boxes + do_nms_sort on the GPU (yl_network_detect_batch /
yl_network_get_boxes_batch / yl_network_get_boxes, csrc/detect.hip) against the oracle's decode +
qsort NMS (common.oracle_boxes = oracle/detect_oracle.c, itself pinned row for row against the
reference's get_network_boxes/do_nms_sort in tests/test_detect_host.py) on the same head tensors.

Bar: rows bit-identical INCLUDING ORDER -- the kernel replays the reference's class-by-class
stable sort, so ties and the order the last class leaves behind must come out the same.
"""
import numpy as np
import pytest

import common
import descs as D
from common import Network

pytestmark = pytest.mark.gpu


def _same_rows(dev: np.ndarray, host: np.ndarray, what):
    assert dev.shape == host.shape, (what, dev.shape, host.shape)
    if not np.array_equal(dev.view(np.uint32), host.view(np.uint32)):
        bad = np.argwhere(dev.view(np.uint32) != host.view(np.uint32))
        r, c = bad[0]
        raise AssertionError("%s: %d cells differ, first at row %d col %d: gpu %r host %r" % (
            what, len(bad), r, c, dev[r, c], host[r, c]))


CASES = [
    # name, width, height, batch, thresh, nms, relative, letter, sizes
    ("yolov3-tiny", 416, 416, 3, 0.10, 0.45, 1, 0, None),
    ("yolov3-tiny", 416, 416, 3, 0.10, 0.45, 0, 0, [(768, 576), (640, 480), (333, 500)]),
    ("yolov3-tiny", 416, 416, 3, 0.10, 0.45, 0, 1, [(768, 576), (640, 480), (333, 500)]),
    ("yolov3-tiny", 416, 416, 2, 0.10, 0.45, 1, 1, (500, 375)),
    ("yolov3", 160, 160, 2, 0.08, 0.40, 1, 0, None),
    ("yolov2-voc", 416, 416, 2, 0.05, 0.40, 1, 0, None),      # region head: 845 detections per image
    ("yolov2-voc", 416, 416, 2, 0.05, 0.40, 0, 1, [(768, 576), (100, 300)]),
    ("tiny-yolo-xnor", 416, 416, 2, 0.05, 0.30, 1, 0, None),
    ("yolov3-tiny", 416, 416, 2, 0.10, 0.0, 1, 0, None),      # nms off: scan order, sort_class 0
]


@pytest.mark.parametrize("name,width,height,batch,thresh,nms,relative,letter,sizes", CASES)
def test_detect_batch_rows_equal_reference_rows(name, width, height, batch, thresh, nms, relative, letter, sizes):
    cfg, wts = common.model_files(name, width, height)
    net = Network.load(cfg, wts, batch, 0, device=0)
    x = common.seeded_input(batch, 3, height, width)
    net.predict(x)                                   # also pulls the heads for the host decode
    rows, counts = net.get_boxes_batch(thresh, nms, cap=2048, sizes=sizes, relative=relative, letter=letter)
    heads = common.OracleHeads(net)
    suppressed = 0
    for b in range(batch):
        if sizes is None:
            w, h = 1, 1
        else:
            w, h = sizes if isinstance(sizes, tuple) else sizes[b]
        host = common.oracle_boxes(heads, b, w, h, thresh, nms=nms, relative=relative, letter=letter)
        assert counts[b] == len(host)
        _same_rows(rows[b], host, (name, "image", b))
        # the single-image entry point (the reference's get_network_boxes + do_nms_sort call) is the same pass
        _same_rows(net.get_boxes(b, w, h, thresh, nms=nms, relative=relative, letter=letter), host, (name, "get_boxes", b))
        if nms > 0:
            plain = common.oracle_boxes(heads, b, w, h, thresh, nms=0.0, relative=relative, letter=letter)
            suppressed += int((plain[:, 6:] > 0).sum() - (host[:, 6:] > 0).sum())
    assert sum(counts) > 0, "threshold too high for the synthetic weights: test is vacuous"
    if nms > 0:
        assert suppressed > 0, "no suppression happened: test does not exercise the NMS"
    net.close()


def _yolo_only_net(B, w, h, n, classes, anchors):
    d = D.yolo(B, w, h, n, classes, n, list(range(n)), np.asarray(anchors, dtype=np.float32))
    net = Network.from_desc([d], B, w, h, n * (classes + 5))
    net.to_device(0)
    return net


def test_ties_and_large_class_follow_the_stable_sort():
    """Saturated logistics give bit-equal probabilities (1.0) on overlapping boxes: which box
    survives then depends on the order earlier classes left behind.  108 positive detections in
    class 0 also drive the m > 64 path."""
    B, w, h, n, classes = 2, 6, 6, 3, 5
    rng = np.random.default_rng(5)
    x = np.zeros((B, n, classes + 5, h, w), dtype=np.float32)
    x[:, :, 0:2] = rng.standard_normal((B, n, 2, h, w)) * 0.3
    x[:, :, 2:4] = rng.standard_normal((B, n, 2, h, w)) * 0.2
    x[:, :, 4] = 40.0                                 # objectness -> exactly 1.0
    x[:, :, 5] = 40.0                                 # class 0: 1.0 everywhere (108 ties)
    x[:, :, 6] = np.where(rng.random((B, n, h, w)) < 0.5, 40.0, -40.0)      # class 1: half tie at 1.0
    x[:, :, 7] = rng.standard_normal((B, n, h, w)) * 2                      # class 2: distinct
    x[:, :, 8] = np.round(rng.standard_normal((B, n, h, w)) * 2)            # class 3: many repeated values
    x[:, :, 9] = -40.0                                # class 4: never passes -> skipped class
    net = _yolo_only_net(B, w, h, n, classes, [2.5, 2.0, 1.5, 3.0, 4.0, 4.0])
    net.predict(x.reshape(B, -1))
    for nms in (0.45, 0.2, 0.8):
        rows, counts = net.get_boxes_batch(0.3, nms, cap=256)
        for b in range(B):
            host = common.oracle_boxes(net, b, 1, 1, 0.3, nms=nms)
            assert counts[b] == len(host) == n * w * h
            _same_rows(rows[b], host, ("ties", nms, b))
    net.close()


def test_zero_objectness_detections_move_to_the_end():
    """thresh < 0 lets objectness == 0 through; do_nms_sort first swaps those to the tail
    (src/box.c:300-309) and never sorts them."""
    B, w, h, n, classes = 1, 5, 4, 2, 3
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((B, n, classes + 5, h, w)) * 2).astype(np.float32)
    obj = x[:, :, 4]
    obj[rng.random(obj.shape) < 0.3] = -200.0          # 1/(1+exp(200)) rounds to 0.0f
    net = _yolo_only_net(B, w, h, n, classes, [2.0, 2.0, 3.0, 1.5])
    net.predict(x.reshape(B, -1))
    rows, counts = net.get_boxes_batch(-1.0, 0.45, cap=64)
    host = common.oracle_boxes(net, 0, 1, 1, -1.0, nms=0.45)
    assert (host[:, 4] == 0).any()
    assert counts[0] == len(host)
    _same_rows(rows[0], host, "zero objectness")
    net.close()


def test_overflowing_cap_reports_the_true_count():
    B, w, h, n, classes = 2, 8, 8, 3, 2
    x = np.full((B, n, classes + 5, h, w), 3.0, dtype=np.float32)
    net = _yolo_only_net(B, w, h, n, classes, [1.0, 1.0, 2.0, 2.0, 3.0, 3.0])
    net.predict(x.reshape(B, -1))
    rows, counts = net.get_boxes_batch(0.2, 0.45, cap=50)
    assert list(counts) == [n * w * h] * B
    assert all(len(r) == 50 for r in rows)
    with pytest.raises(Exception):
        net.get_boxes_batch(0.2, 0.45, cap=8192)       # > YL_DETECT_MAX_CAP
    with pytest.raises(Exception):
        net.get_boxes_batch(0.2, 0.45, cap=64, relative=0)   # absolute boxes need image sizes
    net.close()


def test_detect_batch_is_deterministic_and_independent_of_slot_order():
    """the compaction hands out record slots with atomics; the scan key makes the result
    independent of them: 5 runs, identical bytes."""
    cfg, wts = common.model_files("yolov3-tiny", 416, 416)
    net = Network.load(cfg, wts, 4, 0, device=0)
    x = common.seeded_input(4, 3, 416, 416)
    net.predict(x)
    first, c0 = net.get_boxes_batch(0.1, 0.45, cap=2048)
    for _ in range(4):
        again, c1 = net.get_boxes_batch(0.1, 0.45, cap=2048)
        assert np.array_equal(c0, c1)
        for a, b in zip(first, again):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    net.close()


def test_parallel_and_sequential_suppression_agree_on_a_dense_head():
    """an untrained head at a low threshold: ~1000 boxes per image, dozens of positive classes each
    (every class active, long sorted lists, the m > 64 chunks) -- the (image, class)-parallel path
    and the one-workgroup-per-image path must produce the same bytes, and both the host rows"""
    cfg, wts = common.model_files("yolov3-tiny", 416, 416)
    net = Network.load(cfg, wts, 3, 0, device=0)
    x = common.seeded_input(3, 3, 416, 416)
    net.predict(x)
    net.set_nms_mode(0)
    seq, c0 = net.get_boxes_batch(0.02, 0.45, cap=2048)
    net.set_nms_mode(1)
    par, c1 = net.get_boxes_batch(0.02, 0.45, cap=2048)
    assert np.array_equal(c0, c1) and max(c0) > 300
    for b in range(3):
        assert np.array_equal(seq[b].view(np.uint32), par[b].view(np.uint32)), b
        if c0[b] <= 2048:
            host = common.oracle_boxes(net, b, 1, 1, 0.02, nms=0.45)
            _same_rows(par[b], host, ("dense", b))
    net.close()
