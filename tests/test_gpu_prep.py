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
