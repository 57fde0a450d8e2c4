"""Region layer with a softmax tree (YOLO9000: forward_region_layer_cpu's softmax_tree branch,
src/yolov2_forward_network.c:494-507,:556-562; hierarchical decode in get_region_boxes_cpu :690-712;
read_tree src/additionally.c:1895).  The reference ships no YOLO9000 cfg or tree file, so the fixture is a small
synthetic hierarchy; the oracle restatement is pinned against the reference library on it (CPU), the GPU path
against the oracle (bit-exact rows, softmax within expf ulps)."""
import os

import numpy as np
import pytest

import common
from common import Network, fp, refbind

# 12 classes: three roots, then the children of class 0, 1, 3 and 2 -- read_tree groups [3, 2, 3, 2, 2]
TREE = [("a", -1), ("b", -1), ("c", -1), ("a0", 0), ("a1", 0), ("b0", 1), ("b1", 1), ("b2", 1),
        ("a00", 3), ("a01", 3), ("c0", 2), ("c1", 2)]
GROUPS = [3, 2, 3, 2, 2]

CFG = """[net]
batch=1
subdivisions=1
width=%d
height=%d
channels=3
[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky
[convolutional]
batch_normalize=1
filters=32
size=3
stride=2
pad=1
activation=leaky
[convolutional]
size=1
stride=1
pad=1
filters=51
activation=linear
[region]
anchors = 1,1, 2,3, 4,2
classes=12
coords=4
num=3
softmax=1
tree=%s
"""


def tree_files(width, height):
    from yolo2_light_amd import weights as W
    tree = os.path.join(common.workdir(), "synthetic12.tree")
    with open(tree, "w") as f:
        for name, parent in TREE:
            f.write("%s %d\n" % (name, parent))
    text = CFG % (width, height, tree)
    cfg = os.path.join(common.workdir(), "tree12-%dx%d.cfg" % (width, height))
    open(cfg, "w").write(text)
    wts = cfg[:-4] + ".weights"
    # a wide last layer so that some hierarchical probabilities exceed .5
    W.write_synthetic_weights(text, wts, seed=5)
    return cfg, wts


def test_cfg_parser_reads_the_tree_like_read_tree():
    cfg, wts = tree_files(64, 64)
    net = Network.from_cfg(cfg, 1, 0)
    parent, groups = net.layer_tree(net.n - 1)
    assert list(parent) == [p for _, p in TREE]
    assert list(groups) == GROUPS
    net.close()


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("nms", [0.0, 0.4])
def test_oracle_tree_equals_reference(olib, nms):
    """oracle_region_tree and the oracle's hierarchical decode, bit for bit against the reference library"""
    width, height, batch = 96, 64, 2
    cfg, wts = tree_files(width, height)
    ref = refbind.RefNetwork(cfg, wts, batch, 0)
    x = common.seeded_input(batch, 3, height, width) * 4.0 - 1.5
    ref.predict(x)
    li = ref.layer_info(ref.n - 1)
    assert li["type"] == common.REGION
    conv_out = ref.layer_output(ref.n - 2).copy()
    region_ref = ref.layer_output(ref.n - 1).copy()
    got = np.zeros_like(region_ref)
    gs = np.array(GROUPS, dtype=np.int32)
    import ctypes as C
    olib.oracle_region_tree(fp(conv_out), fp(got), batch, li["n"], li["classes"], li["coords"], li["w"] * li["h"],
                            gs.ctypes.data_as(C.POINTER(C.c_int)), len(gs))
    assert np.array_equal(got.view(np.uint32), region_ref.view(np.uint32))
    # decode: the reference multiplies l.output in place, so every call needs a fresh forward
    net = Network.from_cfg(cfg, batch, 0)                 # host model only: topology + tree for the oracle heads
    nonzero = 0
    for b in range(batch):
        ref.predict(x)
        heads = common.OracleHeads(net, outputs={net.n - 1: ref.layer_output(ref.n - 1).copy()})
        r = ref.get_detections(b, 640, 480, 0.2, nms=nms, relative=0)
        g = common.oracle_boxes(heads, b, 640, 480, 0.2, nms=nms, relative=0)
        assert r.shape == g.shape
        assert np.array_equal(r.view(np.uint32), g.view(np.uint32)), "image %d" % b
        nonzero += int((r[:, 6:] > 0).sum())
    assert nonzero > 0, "fixture: no class survived the hierarchy"
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("batch,nms", [(2, 0.0), (3, 0.4)])
def test_gpu_region_tree_and_hierarchical_decode(olib, batch, nms):
    width, height = 96, 64
    cfg, wts = tree_files(width, height)
    net = Network.load(cfg, wts, batch, 0, device=0)
    x = common.seeded_input(batch, 3, height, width) * 4.0 - 1.5
    net.predict(x)
    li = net.layer_info(net.n - 1)
    conv_out = net.layer_output(net.n - 2)
    got = net.layer_output(net.n - 1)
    ref = np.zeros_like(got)
    gs = np.array(GROUPS, dtype=np.int32)
    import ctypes as C
    olib.oracle_region_tree(fp(conv_out), fp(ref), batch, li["n"], li["classes"], li["coords"], li["w"] * li["h"],
                            gs.ctypes.data_as(C.POINTER(C.c_int)), len(gs))
    g = got.reshape(batch, -1, li["classes"] + 5); r = ref.reshape(batch, -1, li["classes"] + 5)
    assert np.array_equal(g[:, :, :4], r[:, :, :4])                     # flatten: pure indexing
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7)          # expf implementations differ by ulps
    # every group is a probability distribution
    c0 = 5
    for sz in GROUPS:
        np.testing.assert_allclose(g[:, :, c0:c0 + sz].sum(axis=2), 1.0, rtol=1e-5)
        c0 += sz
    # hierarchical decode + NMS on the GPU's own head tensor: rows identical to the oracle's, order included
    heads = common.OracleHeads(net)
    rows, counts = net.get_boxes_batch(0.2, nms, cap=2048, sizes=(640, 480), relative=0, letter=0)
    nonzero = 0
    for b in range(batch):
        host = common.oracle_boxes(heads, b, 640, 480, 0.2, nms=nms, relative=0)
        assert counts[b] == len(host)
        assert np.array_equal(rows[b].view(np.uint32), host.view(np.uint32)), "image %d" % b
        assert ((host[:, 6:] > 0).sum(axis=1) <= 1).all()              # at most one class per box survives
        nonzero += int((host[:, 6:] > 0).sum())
    assert nonzero > 0
    net.close()


def test_tree_cfg_errors_are_reported(tmp_path):
    """a missing tree file, a tree whose size differs from classes=, and map= (the evaluator's class remapping)
    are refused with an error code instead of computing something else"""
    from yolo2_light_amd._lib import lib
    import ctypes as C
    width = height = 64
    tree = str(tmp_path / "t.tree")
    with open(tree, "w") as f:
        for name, parent in TREE[:10]:                      # 10 nodes for classes=12
            f.write("%s %d\n" % (name, parent))
    cases = {
        "missing": (CFG % (width, height, str(tmp_path / "nope.tree")), -2),          # YL_ERR_IO
        "size": (CFG % (width, height, tree), -3),                                      # YL_ERR_CFG
        "map": ((CFG % (width, height, tree)) + "map=%s\n" % tree, -5),                # YL_ERR_UNSUPPORTED
    }
    for what, (text, want) in cases.items():
        cfg = str(tmp_path / (what + ".cfg"))
        open(cfg, "w").write(text)
        h = C.c_void_p()
        rc = lib.yl_network_create_from_cfg(cfg.encode(), 1, 0, C.byref(h))
        assert rc == want, (what, rc, lib.yl_last_error())
        assert lib.yl_last_error()
