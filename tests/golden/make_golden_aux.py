#!/usr/bin/env python
"""Generate tests/golden/aux/*.npz from the REFERENCE ITSELF (oracle/_ref/libyolo2ref.so) for the
neighbours of the hot path:

  front_end.npz   u8 HWC source images + what the reference's load_image (stb PPM decode, /255.)
                  followed by resize_image(im, w, h) returns for them (src/main.c:187-189)
  entropy.npz     float arrays + the multiplier entropy_calibration(x, n, 1/16, 4096) returns
                  (src/yolov2_forward_network_quantized.c:1292)

Run in the build container (where /root/reference exists):  python tests/golden/make_golden_aux.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
from common import refbind  # noqa: E402

FRONT_END = [(37, 23, 64, 96), (100, 60, 96, 96), (9, 1, 32, 32), (1, 7, 32, 32), (64, 48, 64, 48), (90, 120, 64, 64)]
ENTROPY = [(0, 6000, 1.0, "halfnormal"), (1, 5000, 6.0, "halfnormal"), (2, 8000, 0.3, "leaky"), (3, 4000, 30.0, "uniform")]


def entropy_input(seed, n, scale, shape):
    rng = np.random.default_rng(seed)
    if shape == "uniform":
        x = rng.uniform(0, scale, n)
    elif shape == "leaky":
        x = rng.standard_normal(n) * scale
        x = np.where(x > 0, x, 0.1 * x)
    else:
        x = np.abs(rng.standard_normal(n)) * scale
    return x.astype(np.float32)


def main():
    rl = refbind._bind(refbind.GOLD)
    out = {}
    for k, (sw, sh, w, h) in enumerate(FRONT_END):
        rng = np.random.default_rng(1000 + k)
        pix = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
        path = os.path.join(common.workdir(), "aux_%d.ppm" % k)
        common.write_ppm(path, pix)
        ref = np.zeros((3, h, w), dtype=np.float32)
        assert rl.ref_load_resized(path.encode(), w, h, common.fp(ref), None, None) == 0
        out["pix_%d" % k] = pix
        out["ref_%d" % k] = ref
    p = os.path.join(HERE, "aux", "front_end.npz")
    np.savez_compressed(p, n=np.array(len(FRONT_END)), **out)
    print(p, os.path.getsize(p), "bytes")

    out = {}
    for k, (seed, n, scale, shape) in enumerate(ENTROPY):
        x = entropy_input(seed, n, scale, shape)
        out["x_%d" % k] = x
        out["mult_%d" % k] = np.float32(rl.ref_entropy_calibration(common.fp(x), x.size, 1.0 / 16, 4096))
    p = os.path.join(HERE, "aux", "entropy.npz")
    np.savez_compressed(p, n=np.array(len(ENTROPY)), **out)
    print(p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
