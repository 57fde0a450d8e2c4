"""Per-layer agreement of the HIP -quantized path with the reference's network_predict_quantized, end to end (no teacher
forcing): where does the difference at the heads come from?  usage: python tools/diag_int8_e2e.py [model size]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from common import Network  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "yolov3-tiny"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
cfg, wts = common.model_files(name, size, size)
ref = common.refbind.RefNetwork(cfg, wts, 1, 1)
net = Network.load(cfg, wts, 1, 1, device=0)
x = common.seeded_input(1, 3, size, size)
ref.predict(x)
net.predict(x)
for i in range(net.n):
    li = net.layer_info(i)
    g = net.layer_output(i).astype(np.float64)
    r = ref.layer_output(i).astype(np.float64)
    rms = np.sqrt(np.mean(r * r))
    err = np.sqrt(np.mean((g - r) ** 2)) / max(rms, 1e-30)
    nd = int(np.sum(g != r))
    print("%3d type=%2d %-40s rel_rms_err %.3g  differing %d / %d  max|d| %.3g" % (
        i, li["type"], net.layer_kernel(i), err, nd, g.size, np.abs(g - r).max()))
