#!/usr/bin/env python
"""Residual-block micro network for kernel work: [maxpool 1/1 identity] + N x {conv 1x1 C->C/2, conv 3x3 C/2->C,
shortcut from -3} + linear 1x1 head, at a yolov3 stage's geometry, with fusion on -- the producer/consumer
context (int8 side outputs, fused [shortcut]) the convolutions have inside the real network.  Prints the
per-layer device times; run it under rocprofv3 --pmc for per-dispatch counters (tools/pmc_dispatch.py).
  python tools/block_bench.py --mode int8 --C 512 --H 38 --batch 64 [--i8-tile 3] [--tile 0] [--iters 5]"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cfg_text(C, H, blocks, xnor):
    s = "[net]\nbatch=1\nsubdivisions=1\nwidth=%d\nheight=%d\nchannels=%d\n" % (H, H, C)
    s += "[maxpool]\nsize=1\nstride=1\npadding=0\n"
    for _ in range(blocks):
        s += "[convolutional]\nbatch_normalize=1\nfilters=%d\nsize=1\nstride=1\npad=1\nactivation=leaky\n" % (C // 2)
        s += "[convolutional]\n%sbatch_normalize=1\nfilters=%d\nsize=3\nstride=1\npad=1\nactivation=leaky\n" % (
            "xnor=1\n" if xnor else "", C)
        s += "[shortcut]\nfrom=-3\nactivation=linear\n"
    s += "[convolutional]\nfilters=18\nsize=1\nstride=1\npad=1\nactivation=linear\n"
    s += "[yolo]\nmask=0,1,2\nanchors=10,13,16,30,33,23\nclasses=1\nnum=3\n"
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="int8", choices=["fp32", "int8", "xnor"])
    ap.add_argument("--C", type=int, default=512)
    ap.add_argument("--H", type=int, default=38)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--i8-tile", type=int, default=0)
    ap.add_argument("--no-fuse", action="store_true")
    args = ap.parse_args()
    import torch
    from yolo2_light_amd import Network, weights
    work = tempfile.mkdtemp(prefix="yl_block_")
    text = cfg_text(args.C, args.H, args.blocks, args.mode == "xnor")
    cfg = os.path.join(work, "block.cfg")
    open(cfg, "w").write(text)
    wts = os.path.join(work, "block.weights")
    weights.write_synthetic_weights(text, wts, seed=5)
    net = Network.load(cfg, wts, args.batch, 1 if args.mode == "int8" else 0, device=0, fuse=not args.no_fuse)
    if args.tile:
        net.set_conv_tile(args.tile)
    if args.i8_tile:
        net.set_int8_tile(args.i8_tile)
    x = torch.randn((args.batch, args.C, args.H, args.H), device="cuda:0", dtype=torch.float32)
    net.profile(x.data_ptr(), 1)
    ms, tot = net.profile(x.data_ptr(), args.iters)
    B = args.batch
    for i, li in enumerate(net.layers()):
        extra = ""
        if li["type"] == 0:
            fl = 2.0 * li["n"] * li["size"] ** 2 * li["c"] * li["out_h"] * li["out_w"] * B
            extra = "%7.1f Tops/s" % (fl / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0)
        print("%2d type=%2d %-32s %8.4f ms %s" % (i, li["type"], net.layer_kernel(i), ms[i], extra))
    print("total %.4f ms" % tot)


if __name__ == "__main__":
    main()
