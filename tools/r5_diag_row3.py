#!/usr/bin/env python
"""Diagnostic (lab): where K1r's outputs differ from a float64 convolution, per shape and tile."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import descs as D  # noqa: E402
from yolo2_light_amd import Network  # noqa: E402

SHAPES = [
    (3, 32, 19, 19, 70, D.LINEAR), (3, 32, 19, 19, 70, D.LEAKY), (1, 32, 19, 19, 70, D.LINEAR), (3, 32, 19, 19, 128, D.LINEAR),
    (3, 32, 20, 20, 70, D.LINEAR), (3, 16, 19, 19, 70, D.LINEAR), (1, 256, 13, 13, 512, D.LEAKY), (2, 16, 13, 13, 33, D.LEAKY),
    (1, 256, 13, 13, 128, D.LEAKY), (1, 64, 13, 13, 128, D.LEAKY), (2, 64, 38, 38, 128, D.LEAKY),
]


def main():
    tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "61,62,68").split(",")]
    for shape in SHAPES:
        B, Cc, H, W, M, act = shape
        rng = np.random.default_rng(2718 + M + H)
        K = Cc * 9
        wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
        bias = rng.normal(0, 0.5, M).astype(np.float32)
        x = (rng.standard_normal((B, Cc, H, W)) * np.exp(rng.uniform(-6, 3, (B, Cc, H, W)))).astype(np.float32)
        truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wts.reshape(M, Cc, 3, 3)).double(),
                                           torch.from_numpy(bias).double(), stride=1, padding=1)
        if act == D.LEAKY:
            truth = torch.where(truth > 0, truth, 0.1 * truth)
        truth = truth.numpy()
        rms = float(np.sqrt(np.mean(truth ** 2)))
        d = D.conv(B, W, H, Cc, M, 3, 1, 1, act, wts, bias)
        net = Network.from_desc([d], B, W, H, Cc, 0)
        net.set_variant(0)
        net.to_device(0)
        for t in tiles:
            net.set_conv_tile(t)
            outs = [net.predict(x).copy().reshape(B, M, H, W) for _ in range(2)]
            got = outs[0].astype(np.float64)
            bad = ~(np.abs(got - truth) <= 1e-3 * rms)          # NaN counts as bad
            same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
            print("shape %r tile %d %s: bad %d of %d, nan %d, repeatable %s" % (shape, t, net.layer_kernel(0), int(bad.sum()), bad.size,
                                                                               int(np.isnan(got).sum()), same), flush=True)
            if bad.any():
                idx = np.argwhere(bad)
                print("   first bad (b, m, oy, ox):", [tuple(int(v) for v in r) for r in idx[:12]])
                for ax, nm in enumerate(("b", "m", "oy", "ox")):
                    vals, cnt = np.unique(idx[:, ax], return_counts=True)
                    print("   by %s:" % nm, dict(zip(vals.tolist()[:40], cnt.tolist()[:40])))
                k = tuple(idx[0])
                print("   got %r truth %r" % (outs[0][k], truth[k]))
        net.close()


if __name__ == "__main__":
    main()
