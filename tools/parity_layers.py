#!/usr/bin/env python
"""Per-layer fp32_close ratio / strict max-rel of the HIP path against the reference library (oracle/_ref) on
yolov3-608 batch 1, for a list of schedule/kernel variants.  Usage: python tools/parity_layers.py 0 30"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from common import Network, refbind, fp32_close

name, width, height, batch = "yolov3", 608, 608, 1
cfg, wts = common.model_files(name, width, height)
ref = refbind.RefNetwork(cfg, wts, batch, 0)
x = common.seeded_input(batch, 3, height, width)
ref.predict(x)
want = [ref.layer_output(i).copy() for i in range(ref.n)]
for v in [int(a) for a in sys.argv[1:]] or [0, 30]:
    net = Network.load(cfg, wts, batch, 0, device=0)
    net.set_variant(v)
    net.predict(x)
    rows = []
    for i in range(net.n):
        ok, ratio, worst = fp32_close(net.layer_output(i), want[i])
        rows.append((ratio, i, net.layer_kernel(i), common.strict_max_rel(net.layer_output(i), want[i])))
    print("variant %d: worst ratio %.3f" % (v, max(r[0] for r in rows)))
    for r in sorted(rows, reverse=True)[:8]:
        print("   layer %3d %-40s ratio %.3f strict %.2e" % (r[1], r[2], r[0], r[3]))
    print("   first layers:", " ".join("%d:%.3f" % (r[1], r[0]) for r in rows[:12]))
    net.close()
