"""End-to-end drop-in: the REFERENCE's own host code (parse_network_cfg,
load_weights_upto_cpu, fuse/binarize/quantize prep, get_network_boxes,
do_nms_sort -- compiled unmodified into oracle/_ref/libyolo2ref_hip.so) calls
`network_predict_hip` from integration/network_predict_hip.c (the binding
INTEGRATION.md tells a maintainer to add), which reaches the HIP kernels only
through the C-ABI.  Results are compared with the reference CPU path run on the
very same `network` object."""
import os

import numpy as np
import pytest

import common
from common import Network, refbind, fp32_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_drop_in_library():
    common.require_ref(hip=True)        # a failure on a GPU box, a skip only without a GPU


def _heads(ref):
    return [i for i in range(ref.n) if ref.layer_info(i)["type"] in (common.YOLO, common.REGION)]


@pytest.mark.parametrize("name,width,height,batch", [("yolov3-tiny", 416, 416, 2), ("yolov3", 160, 160, 2),
                                                     ("tiny-yolo-xnor", 416, 416, 1)])
def test_reference_host_code_drives_hip_path(name, width, height, batch):
    cfg, wts = common.model_files(name, width, height)
    ref = refbind.RefNetwork(cfg, wts, batch, 0, hip=True)
    x = common.seeded_input(batch, 3, height, width)
    ref.predict(x)                                            # network_predict_cpu
    heads = _heads(ref)
    cpu_heads = {i: ref.layer_output(i) for i in heads}
    cpu_dets = [ref.get_detections(b, width, height, 0.24, nms=0.4) for b in range(batch)]
    ref.predict_hip(x)                                        # network_predict_hip -> same l.output buffers
    xnor = name == "tiny-yolo-xnor"
    for i in heads:
        got, want = ref.layer_output(i), cpu_heads[i]
        if xnor:
            # a sign flip of a near-zero FP32 activation changes a count by 1: statistical agreement
            bad = np.abs(got - want) > (1e-3 + 1e-3 * np.abs(want))
            assert bad.mean() < 0.02
        else:
            ok, ratio, worst = fp32_close(got, want)
            assert ok, "head layer %d: err/allowed %.3g" % (i, ratio)
    for b in range(batch):
        g = ref.get_detections(b, width, height, 0.24, nms=0.4)   # the reference's own decode + NMS
        r = cpu_dets[b]
        assert abs(len(g) - len(r)) <= max(2, len(r) // 20)
        if len(r) and len(g) and not xnor:
            with np.errstate(invalid="ignore", over="ignore"):
                dist = (np.abs(r[:, None, :4] - g[None, :, :4]) / (1e-5 + 1e-4 * np.abs(r[:, None, :4]))).max(axis=2)
            j = np.nan_to_num(dist, nan=0.0).argmin(axis=1)
            assert (np.nan_to_num(dist, nan=0.0)[np.arange(len(r)), j] < 1.0).mean() > 0.98
    ref.lib.ref_free_hip()


def test_reference_host_code_quantized():
    name, width, height = "yolov3-tiny", 416, 416
    cfg, wts = common.model_files(name, width, height)
    ref = refbind.RefNetwork(cfg, wts, 1, 1, hip=True)        # -quantized
    x = common.seeded_input(1, 3, height, width)
    ref.predict(x)                                            # network_predict_quantized
    heads = _heads(ref)
    cpu_heads = {i: ref.layer_output(i).astype(np.float64) for i in heads}
    ref.predict_hip(x)
    for i in heads:
        g = ref.layer_output(i).astype(np.float64); r = cpu_heads[i]
        # golden (scalar) build of the reference: its first FP32 layer rounds products and sums separately, ours is the fused chain
        # of its AVX build -- a few dozen int8 codes of the next layer flip and the synthetic weights amplify that to ~5e-3 here.
        # The tight end-to-end pin (1e-4 against the AVX build, flip fraction against this one) is
        # tests/test_gpu_int8_xnor.py::test_int8_network_vs_reference_library_batch1.
        err = np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r * r))
        print("quantized drop-in head %d: relative RMS error vs network_predict_quantized (scalar build) %.3g" % (i, err))
        assert err < 0.05
    ref.lib.ref_free_hip()


def test_reference_host_code_softmax_tree():
    """A [region] layer with tree= (YOLO9000-style hierarchy): the reference's parser reads the tree, the adaptor
    hands l.softmax_tree's parent / group arrays through yl_layer_desc, the head tensor matches the CPU path."""
    import test_softmax_tree as T
    width, height, batch = 96, 64, 2
    cfg, wts = T.tree_files(width, height)
    ref = refbind.RefNetwork(cfg, wts, batch, 0, hip=True)
    x = common.seeded_input(batch, 3, height, width) * 4.0 - 1.5
    ref.predict(x)
    want = ref.layer_output(ref.n - 1).copy()
    ref.predict_hip(x)
    got = ref.layer_output(ref.n - 1)
    sz = 12 + 5
    assert np.abs(got.reshape(-1, sz)[:, 5:8].sum(axis=1) - 1.0).max() < 1e-5        # the root group is a softmax
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-6)
    ref.lib.ref_free_hip()


@pytest.mark.parametrize("name,size,batch,quantized", [("yolov3-tiny", 96, 5, 0), ("yolov3", 64, 6, 0), ("yolov3", 64, 5, 1),
                                                       ("tiny-yolo-xnor", 96, 7, 0)])
def test_pipelined_predict_equals_one_pass(name, size, batch, quantized):
    """yl_network_predict runs the batch as sub-batches (input of k+1 and heads of k-1 on the copy streams while k computes,
    runtime.hip BatchWindow): every split, uneven ones included, leaves the bits of ONE pass over the whole batch in every
    tensor and in the returned host heads."""
    import torch
    cfg, wts = common.model_files(name, size, size)
    x = common.seeded_input(batch, 3, size, size)
    net = Network.load(cfg, wts, batch, quantized, device=0, fuse=True)
    xd = torch.from_numpy(x).to("cuda:0")
    net.forward_device(xd.data_ptr())
    net.synchronize()
    want = {i: net.layer_output(i).copy() for i in range(net.n) if net.layer_materialised(i)}
    old = os.environ.get("YL_PREDICT_SPLIT")
    try:
        for split in (1, 2, 3, 4):
            os.environ["YL_PREDICT_SPLIT"] = str(split)
            xd.zero_()
            last = net.predict(x)
            for i, w in want.items():
                assert np.array_equal(net.layer_output(i).view(np.uint32), w.view(np.uint32)), "split %d layer %d" % (split, i)
            if last is not None:
                assert np.array_equal(np.asarray(last).ravel().view(np.uint32), want[net.n - 1].ravel().view(np.uint32))
    finally:
        if old is None:
            os.environ.pop("YL_PREDICT_SPLIT", None)
        else:
            os.environ["YL_PREDICT_SPLIT"] = old
    net.close()
