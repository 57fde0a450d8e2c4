#!/usr/bin/env python
"""Per-layer FP32 error of yolov3-608 batch 1 against a FLOAT64 ground truth (common.TruthNet), for
  * the reference's scalar build   (oracle/_ref/libyolo2ref.so,      -O2, gemm_nn as written)
  * the reference's AVX build      (oracle/_ref/libyolo2ref_fast.so, `make AVX=1 OPENMP=1`, -Ofast)
  * the HIP path, Winograd on (default) and off
plus the same comparison on the detections.  VERDICT round 2, next-round item 2: the numbers behind the FP32
contract, measured instead of asserted.   Usage: python tools/parity_layers.py [name width height]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from common import Network, refbind, TruthNet, error_vs_truth


def main():
    name, width, height = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("yolov3", 608, 608)
    batch = 1
    cfg, wts = common.model_files(name, width, height)
    x = common.seeded_input(batch, 3, height, width)
    t0 = time.time()
    host = Network.load(cfg, wts, batch, 0)
    truth = TruthNet(host, open(cfg).read())
    truth.forward(x)
    print("# float64 truth: %.1f s" % (time.time() - t0))
    runs = {}
    t0 = time.time()
    ref = refbind.RefNetwork(cfg, wts, batch, 0)
    ref.predict(x)
    runs["ref_scalar"] = [ref.layer_output(i) for i in range(ref.n)]
    print("# reference scalar: %.1f s" % (time.time() - t0))
    if refbind.available(fast=True):
        t0 = time.time()
        fast = refbind.RefNetwork(cfg, wts, batch, 0, fast=True)
        fast.predict(x)
        runs["ref_avx"] = [fast.layer_output(i) for i in range(fast.n)]
        print("# reference AVX+OpenMP: %.1f s" % (time.time() - t0))
    kernels = {}
    for tag, wino in (("hip", True), ("hip_nowino", False)):
        net = Network.load(cfg, wts, batch, 0, device=0, winograd=wino)
        net.predict(x)
        runs[tag] = [net.layer_output(i) for i in range(net.n)]
        kernels[tag] = [net.layer_kernel(i) for i in range(net.n)]
        net.close()
    tags = list(runs)
    print("# per layer: relative RMS error | max error / RMS(truth), against the float64 truth")
    print("%5s %-44s" % ("layer", "kernel") + "".join(" %21s" % t for t in tags))
    worst = {t: [0.0, 0.0] for t in tags}
    for i in range(host.n):
        row = "%5d %-44s" % (i, kernels["hip"][i][:44])
        for t in tags:
            e = error_vs_truth(runs[t][i], truth.outputs[i])
            worst[t][0] = max(worst[t][0], e[0]); worst[t][1] = max(worst[t][1], e[1])
            row += "  %9.3g | %8.3g" % e
        print(row)
    print("%5s %-44s" % ("worst", "") + "".join("  %9.3g | %8.3g" % tuple(worst[t]) for t in tags))
    # heads: pure relative error where the truth is not a cancellation result
    heads = [i for i, li in enumerate(host.layers()) if li["type"] == common.YOLO]
    for i in heads:
        t = truth.outputs[i]
        print("head %d: fraction of elements within 1e-4 relative of the truth: " % i +
              "  ".join("%s %.5f" % (tg, float(np.mean(np.abs(runs[tg][i] - t) <= 1e-4 * np.abs(t)))) for tg in tags))
        s = runs["ref_scalar"][i].astype(np.float64)
        print("head %d: fraction differing from the reference scalar build by more than 1e-4 relative: " % i +
              "  ".join("%s %.2e" % (tg, float(np.mean(np.abs(runs[tg][i] - s) > 1e-4 * np.abs(s)))) for tg in tags if tg != "ref_scalar"))


if __name__ == "__main__":
    main()
