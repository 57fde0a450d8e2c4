#!/usr/bin/env python
"""Tile sweep of the K1 FP32 MFMA conv over the unique conv shapes of yolov3-608
 at batch B: every shape x every tile config, HIP-event timed through
yl_network_profile.  Prints one JSON line per (shape, tile) and a best-tile table.

    python tools/sweep_conv.py --batch 64 --tiles 1,2,3,6,7,8 --iters 3
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (count, M, C, size, stride, H(=W) of the input) for yolov3 at 608x608
SHAPES_608 = [
    (1, 32, 3, 3, 1, 608), (1, 64, 32, 3, 2, 608), (1, 32, 64, 1, 1, 304), (1, 64, 32, 3, 1, 304),
    (1, 128, 64, 3, 2, 304), (2, 64, 128, 1, 1, 152), (2, 128, 64, 3, 1, 152), (1, 256, 128, 3, 2, 152),
    (10, 128, 256, 1, 1, 76), (11, 256, 128, 3, 1, 76), (1, 512, 256, 3, 2, 76), (10, 256, 512, 1, 1, 38),
    (11, 512, 256, 3, 1, 38), (1, 1024, 512, 3, 2, 38), (7, 512, 1024, 1, 1, 19), (7, 1024, 512, 3, 1, 19),
    (1, 255, 1024, 1, 1, 19), (1, 256, 512, 1, 1, 19), (1, 256, 768, 1, 1, 38), (1, 255, 512, 1, 1, 38),
    (1, 128, 256, 1, 1, 38), (1, 128, 384, 1, 1, 76), (1, 255, 256, 1, 1, 76),
]


# the 3x3 / stride-1 layers of yolov3-tiny at 416x416 that K1r takes (no fused [maxpool]) and its 1x1 layers
SHAPES_TINY_416 = [
    (1, 512, 256, 3, 1, 13), (1, 1024, 512, 3, 1, 13), (1, 512, 256, 3, 1, 13), (1, 256, 384, 3, 1, 26),
    (1, 256, 1024, 1, 1, 13), (1, 255, 512, 1, 1, 13), (1, 128, 256, 1, 1, 13), (1, 255, 256, 1, 1, 26),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="yolov3-608", choices=["yolov3-608", "yolov3-tiny-416"], help="which network's conv shapes")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tiles", default="0,12,14,20,22,31",
                    help="forced tile ids: 11..22 = direct kernel tiles, 0 = heuristic, 31 = Winograd (3x3/1/1 only)")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="", help="comma list of shape indices")
    ap.add_argument("--external-input", action="store_true", help="time the layer on a caller-owned device tensor (no front pad)")
    ap.add_argument("--variant", type=int, default=-1, help="yl_network_set_variant bits, applied before to_device")
    ap.add_argument("--input", default="rand", choices=["rand", "zeros", "relu"],
                    help="input data: U(-0.3, 0.7), all zeros (DVFS probe: same instructions, less switching), or a leaky-like mix")
    ap.add_argument("--weights", default="rand", choices=["rand", "zeros"])
    args = ap.parse_args()
    import torch
    import descs as D
    from yolo2_light_amd import Network

    tiles = [int(t) for t in args.tiles.split(",")]
    shapes = SHAPES_608 if args.set == "yolov3-608" else SHAPES_TINY_416
    only = [int(i) for i in args.only.split(",")] if args.only else range(len(shapes))
    rng = np.random.default_rng(0)
    B = args.batch
    total_best = 0.0
    total_flops = 0.0
    for si in only:
        cnt, M, Cc, size, stride, H = shapes[si]
        pad = size // 2
        K = Cc * size * size
        wts = rng.normal(0, np.sqrt(2.0 / K), M * K).astype(np.float32)
        if args.weights == "zeros":
            wts[:] = 0
        bias = rng.normal(0, 0.1, M).astype(np.float32)
        d = D.conv(B, H, H, Cc, M, size, stride, pad, D.LEAKY, wts, bias)
        net = Network.from_desc([d], B, H, H, Cc)
        if args.variant >= 0:
            net.set_variant(args.variant)
        net.to_device(0)
        x = torch.rand((B, Cc, H, H), device="cuda:0", dtype=torch.float32) - 0.3
        if args.input == "zeros":
            x.zero_()
        elif args.input == "relu":
            x = torch.where(x > 0, x, 0.1 * x)
        # run from the library's own input tensor (front-padded like every activation tensor inside a network, which is
        # what selects the Winograd kernel's folded-mask transform and its 64 x 32 tiling), not from torch's buffer
        net.predict(x.cpu().numpy())
        xin = net.input_dev if not args.external_input else x.data_ptr()
        flops = 2.0 * M * K * d.out_h * d.out_w * B
        best = None
        for t in tiles:
            net.set_conv_tile(t)
            try:
                net.profile(xin, 1)          # warm-up
                ms, _ = net.profile(xin, args.iters)
            except Exception as e:                    # a tile that does not apply to this shape
                print("# shape %d tile %d: %s" % (si, t, e), flush=True)
                continue
            finally:
                net.set_conv_tile(0)
            tf = flops / (ms[0] * 1e-3) / 1e12
            rec = {"shape": si, "M": M, "C": Cc, "size": size, "stride": stride, "H": H, "count": cnt,
                   "tile": t, "kernel": net.layer_kernel(0), "ms": float(ms[0]), "tflops": float(tf)}
            print(json.dumps(rec), flush=True)
            if t != 0 and (best is None or ms[0] < best[1]):
                best = (t, float(ms[0]), tf, net.layer_kernel(0))
        if best:
            total_best += best[1] * cnt
            total_flops += flops * cnt
            print("# shape %2d x%-2d M=%4d C=%4d k=%d s=%d H=%3d  best tile %d %-28s %.3f ms %.1f TF" % (
                si, cnt, M, Cc, size, stride, H, best[0], best[3], best[1], best[2]), flush=True)
        net.close()
        del x
        torch.cuda.empty_cache()
    if total_best > 0:
        print("# all conv layers with per-shape best tile: %.2f ms per batch of %d -> %.1f TF, %.1f img/s (conv only)" % (
            total_best, B, total_flops / (total_best * 1e-3) / 1e12, B / (total_best * 1e-3)))


if __name__ == "__main__":
    main()
