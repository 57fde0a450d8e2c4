#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libyolo2ref.so,
the unmodified reference CPU path built by oracle/Makefile with the golden flags).

Run in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
Each fixture holds, for one (model, size, batch, mode) case on the deterministic
synthetic weights + seeded input: sha256 of the weights file, float64 sum / abs-sum
of EVERY layer output, the full tensors of the detection heads and of the conv
feeding each head, and the detections of main.c:228-229 (thresh .24, nms .4).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
from common import refbind  # noqa: E402

CASES = [
    ("yolov3-tiny", 96, 96, 2, 0),
    ("yolov3-tiny", 96, 96, 1, 1),
    ("yolov3", 64, 64, 1, 0),
    ("yolov3", 64, 64, 1, 1),
    ("tiny-yolo-xnor", 96, 96, 2, 0),
]


def case_name(name, w, h, b, q):
    return "%s_%dx%d_b%d_%s" % (name, w, h, b, "int8" if q else "fp32")


def main():
    for name, w, h, b, q in CASES:
        cfg, wts = common.model_files(name, w, h)
        ref = refbind.RefNetwork(cfg, wts, b, q)
        x = common.seeded_input(b, 3, h, w)
        ref.predict(x)
        sums = np.zeros((ref.n, 2), np.float64)
        keep = {}
        for i in range(ref.n):
            o = ref.layer_output(i)
            if q:
                o = o.reshape(b, -1)[:1]         # the reference's INT8 path computes batch item 0 only
            sums[i] = (o.astype(np.float64).sum(), np.abs(o.astype(np.float64)).sum())
            li = ref.layer_info(i)
            if li["type"] in (common.YOLO, common.REGION):
                keep["layer_%d" % i] = ref.layer_output(i)
                keep["layer_%d" % (i - 1)] = ref.layer_output(i - 1)
        dets = [ref.get_detections(bi, w, h, 0.24, nms=0.4) for bi in range(b if not q else 1)]
        sha = hashlib.sha256(open(wts, "rb").read()).hexdigest()
        out = os.path.join(HERE, case_name(name, w, h, b, q) + ".npz")
        np.savez_compressed(out, weights_sha256=np.array(sha), input_sha256=np.array(hashlib.sha256(x.tobytes()).hexdigest()),
                            layer_sums=sums, n_dets=np.array([len(d) for d in dets]),
                            **{"dets_%d" % i: d for i, d in enumerate(dets)}, **keep)
        print(out, os.path.getsize(out), "bytes,", [len(d) for d in dets], "detections")


if __name__ == "__main__":
    main()
