"""yl_group_* (csrc/group.hip): the single-process multi-GPU form of the path on however many GPUs the box
has (the GPU test box has one: the group of 1 still runs the worker thread, the replica built from a host
model and the RCCL communicator + send/recv gather).  Results must equal the plain single-network path
bit for bit."""
import numpy as np
import pytest

import common
from common import Network
from yolo2_light_amd import parallel
from yolo2_light_amd._lib import lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quantized", [0, 1])
def test_group_equals_single_network(quantized):
    name, width, height, batch = "yolov3-tiny", 160, 160, 5
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    single = Network.load(cfg, wts, batch, quantized, device=0, fuse=True)
    out1 = single.predict(x)
    rows1, counts1 = single.get_boxes_batch(0.1, 0.45, cap=512)
    ndev = lib.yl_device_count()
    for devices in ([0], list(range(min(ndev, 4)))):
        model = Network.load(cfg, wts, batch, quantized, fuse=True)        # host model, not on a device
        grp = parallel.Group(model, devices)
        model.close()                                                      # the group owns its replicas
        spans = [grp.shard(r) for r in range(grp.n)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == batch
        out = grp.predict(x)
        assert np.array_equal(out.view(np.uint32), out1.view(np.uint32))
        rows, counts = grp.get_boxes_batch(0.1, 0.45, cap=512)             # through RCCL send/recv
        assert np.array_equal(counts, counts1) and counts.sum() > 0
        for b in range(batch):
            assert np.array_equal(rows[b].view(np.uint32), rows1[b].view(np.uint32)), b
        # per-layer readback through a member replica
        m0 = grp.member(0)
        f0, c0 = grp.shard(0)
        last = single.n - 1
        assert np.array_equal(m0.layer_output(last), single.layer_output(last).reshape(batch, -1)[f0:f0 + c0].reshape(-1))
        # asynchronous form: device inputs already resident on each rank
        grp.forward(None)
        grp.synchronize()
        grp.close()
    single.close()
