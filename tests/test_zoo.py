"""The generated cfg text builds the same network as the cfg files the reference
ships -- checked with the REFERENCE's parser, only where the reference tree is
present (build container)."""
import os

import pytest

import common
from common import refbind
from yolo2_light_amd import zoo

REF_BIN = "/root/reference/bin"
PAIRS = [("yolov3-tiny", "yolov3-tiny.cfg", 416), ("yolov3", "yolov3.cfg", 416), ("tiny-yolo-xnor", "tiny-yolo-obj_xnor.cfg", 416),
         ("yolov3-spp", "yolov3-spp.cfg", 608), ("yolov3-openimages", "yolov3-openimages.cfg", 608),
         ("yolov2-voc", "yolov2-voc.cfg", 416), ("tiny-yolo-voc", "tiny-yolo-voc.cfg", 416)]


@pytest.mark.skipif(not (os.path.isdir(REF_BIN) and refbind.available()), reason="reference tree not present")
@pytest.mark.parametrize("name,ref_file,size", PAIRS)
@pytest.mark.parametrize("quantized", [0, 1])
def test_generated_cfg_equals_shipped_cfg(name, ref_file, size, quantized):
    ours = zoo.write_cfg(name, common.workdir(), size, size)
    a = refbind.RefNetwork(ours, "", 1, quantized)
    b = refbind.RefNetwork(os.path.join(REF_BIN, ref_file), "", 1, quantized)
    assert a.n == b.n
    for i in range(a.n):
        ia, ib = a.layer_info(i), b.layer_info(i)
        assert ia == ib, (i, ia, ib)
        if quantized and ia["type"] == 0:
            assert a.layer_quant_multipliers(i)[0] == b.layer_quant_multipliers(i)[0], i   # input_calibration list


def test_conv_shapes_walker_counts():
    for name, size, n_conv, gflop in [("yolov3-tiny", 416, 13, 5.565), ("yolov3", 608, 75, 140.69), ("tiny-yolo-xnor", 416, 9, 6.947)]:
        text = zoo.MODELS[name](size, size)
        shapes = zoo.conv_shapes(text)
        assert len(shapes) == n_conv
        path = zoo.write_cfg(name, common.workdir(), size, size)
        from yolo2_light_amd import Network
        net = Network.from_cfg(path, 1, 0)
        assert abs(net.flops_per_image / 1e9 - gflop) < 0.01      # SURVEY 8d / BASELINE.md table
