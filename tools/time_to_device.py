#!/usr/bin/env python
"""Host time of yl_network_to_device for yolov3-608 with the kernel-layout weight images packed by host loops
(device_pack=False: the round-2 path) and by csrc/pack.hip's kernels (device_pack=True, default), FP32 and -quantized.
SURVEY 8f-3 / VERDICT round 2 item 5: "to_device host time measured before/after"."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from common import Network

name, width, height, batch = "yolov3", 608, 608, 1
cfg, wts = common.model_files(name, width, height)
for quantized in (0, 1):
    for dp in (False, True, False, True):
        net = Network.load(cfg, wts, batch, quantized, device_pack=dp)      # host model: parse + load + fuse (+ quantise)
        t0 = time.perf_counter()
        net.to_device(0)
        dt = time.perf_counter() - t0
        print("yolov3-608 batch %d %s  device_pack=%-5s  to_device %.3f s" % (batch, "-quantized" if quantized else "FP32      ", dp, dt), flush=True)
        net.close()
