"""the one-time prep passes on the GPU (csrc/prep.hip, yl_network_prepare_on_device) against the host
passes of host_prep.cpp, which tests/test_host_prep.py pins against the reference library: folded weights and biases,
mean_arr, weights_int8 and both multipliers must be identical bit for bit, and so must a forward pass."""
import numpy as np
import pytest

import common
from common import Network

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,width,height,quantized", [
    ("yolov3-tiny", 96, 96, 0), ("yolov3-tiny", 96, 96, 1), ("yolov3", 64, 64, 1), ("tiny-yolo-xnor", 96, 96, 0),
    ("yolov2-voc", 96, 96, 1),
])
def test_device_prep_equals_host_prep(name, width, height, quantized):
    cfg, wts = common.model_files(name, width, height)
    host = Network.load(cfg, wts, 2, quantized)
    dev = Network.load(cfg, wts, 2, quantized, device_prep=True)
    n_conv = 0
    for i, li in enumerate(host.layers()):
        if li["type"] != common.CONV:
            continue
        n_conv += 1
        assert dev.layer_info(i)["batch_normalize"] == 0 == li["batch_normalize"]
        for what, a, b in (("weights", host.layer_weights(i), dev.layer_weights(i)),
                           ("biases", host.layer_biases(i), dev.layer_biases(i))):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "layer %d %s" % (i, what)
        if li["xnor"]:
            a, b = host.layer_mean_arr(i), dev.layer_mean_arr(i)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "layer %d mean_arr" % i
        if quantized:
            assert np.array_equal(host.layer_weights_int8(i), dev.layer_weights_int8(i)), "layer %d weights_int8" % i
            assert host.layer_quant_multipliers(i) == dev.layer_quant_multipliers(i), "layer %d multipliers" % i
    assert n_conv > 5
    x = common.seeded_input(2, 3, height, width)
    host.to_device(0)
    dev.to_device(0)
    a, b = host.predict(x), dev.predict(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    host.close(); dev.close()


@pytest.mark.parametrize("name,width,height,quantized,bf16,variant", [
    ("yolov3", 64, 64, 0, False, None),            # k-major FP32 panels (both K orders) + Winograd U, default packing
    ("yolov3-tiny", 96, 96, 1, False, None),       # int8 units
    ("yolov3", 64, 64, 0, True, None),             # bf16 units
    ("tiny-yolo-xnor", 96, 96, 0, False, None),    # XNOR sign words (c % 64 != 0 included)
    ("yolov2-voc", 96, 96, 1, False, None),
])
def test_device_packers_equal_host_packers(name, width, height, quantized, bf16, variant):
    """csrc/pack.hip against the host loops of runtime.hip / conv_f32_wino32.hip: every packed weight image a network
    uploads (k-major FP32 panels, Winograd U, int8 / bf16 16-byte units, XNOR sign words, K1x's three-piece bf16 weights, K1r's row-transformed ones)
    identical byte for byte,
    and the same forward pass."""
    cfg, wts = common.model_files(name, width, height)
    nets = [Network.load(cfg, wts, 2, quantized, device=0, bf16=bf16, variant=variant, device_pack=dp) for dp in (False, True)]
    seen = set()
    for i, li in enumerate(nets[0].layers()):
        if li["type"] != common.CONV:
            continue
        for which in (0, 1, 2, 3, 7, 8):       # 7 / 8: the three-piece bf16 weights of K1x / K1r (conv_f32_x3.hip, conv_f32_row3.hip)
            a, b = nets[0].layer_packed(i, which), nets[1].layer_packed(i, which)
            assert (a is None) == (b is None), "layer %d image %d" % (i, which)
            if a is not None:
                seen.add(which)
                assert a.size == b.size and np.array_equal(a, b), "layer %d image %d: %d of %d bytes differ" % (
                    i, which, int((a != b).sum()) if a.size == b.size else -1, a.size)
    assert seen, "no packed image compared"
    x = common.seeded_input(2, 3, height, width)
    a, b = nets[0].predict(x), nets[1].predict(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for n in nets:
        n.close()


def test_device_packers_xnor_fallback():
    """+-mean weights of an xnor convolution outside the bit path (1x1 / stride 2): pack_kmajor's mean branch"""
    import test_gpu_int8_xnor as T
    cfg, wts = T._mixed_xnor_files(64, 48)
    nets = [Network.load(cfg, wts, 1, 0, device=0, device_pack=dp) for dp in (False, True)]
    n_fallback = 0
    for i, li in enumerate(nets[0].layers()):
        if li["type"] == common.CONV and li["xnor"] and li["conv_mode"] == common.CONV_F32:
            n_fallback += 1
            assert np.array_equal(nets[0].layer_packed(i, 0), nets[1].layer_packed(i, 0)), "layer %d" % i
    assert n_fallback >= 1
    for n in nets:
        n.close()
