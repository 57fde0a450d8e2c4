#!/usr/bin/env python
"""BASELINE config 1 ("yolov3-tiny.cfg 416x416 batch=1 on dog.jpg via the CPU path") as a fixture.

Run in the build container (where /root/reference exists):
    python tests/golden/make_golden_dog.py
Everything is produced by the REFERENCE ITSELF (oracle/_ref/libyolo2ref.so = its unmodified sources):
  pixels        the photo as the reference's stb decoder delivers it (load_image, src/additionally.c:3068),
                HWC u8 -- (float)u8/255. reproduces load_image's tensor exactly (checked here)
  sized_sha256  sha256 of resize_image(im, 416, 416) (src/main.c:187-189), the tensor network_predict sees
  fp32 / int8   network_predict_cpu / network_predict_quantized on that tensor with the deterministic
                synthetic weights (no trained weights exist offline): float64 sum / abs-sum of every layer
                output, the two [yolo] head tensors of the FP32 path element by element, and the detections of
                src/main.c:228-229 (thresh .24, nms .4)

`python tests/golden/make_golden_dog.py xnor` writes the second fixture, dog_tiny-yolo-xnor_416.npz: the photo through
bin/tiny-yolo-obj_xnor.cfg's topology (BASELINE config 5's network) on the reference CPU path
(src/yolov2_forward_network.c:116-203, the XNOR branch).  north_star wants the XNOR work BIT-exact on dog.jpg, and
the FP32 first layer in front of it is not (summation order), so besides layer sums and detections the fixture holds,
for each of the 7 XNOR convolutions, the sign bits (x > 0, src/additionally.c:132) of the tensor the REFERENCE fed
it and the sha256 of the tensor the reference got out: a GPU run of that layer on those bits must reproduce the
hash (tests/test_gpu_headline.py::test_dog_jpg_xnor_layers_bit_exact_against_the_reference_fixture).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
from common import refbind  # noqa: E402

DOG = "/root/reference/bin/dog.jpg"
NAME, W, H = "yolov3-tiny", 416, 416
LOW_THRESH = 0.1


def load_resized(lib, path, w, h):
    out = np.zeros((3, h, w), np.float32)
    sw, sh = C.c_int(0), C.c_int(0)
    assert lib.ref_load_resized(path.encode(), w, h, common.fp(out), C.byref(sw), C.byref(sh)) == 0
    return out, sw.value, sh.value


def main():
    lib = refbind._bind(refbind.GOLD)
    sized, sw, sh = load_resized(lib, DOG, W, H)
    full, _, _ = load_resized(lib, DOG, sw, sh)            # resize to its own size = identity
    pixels = np.rint(full.astype(np.float64) * 255.0).astype(np.uint8).transpose(1, 2, 0).copy()     # HWC
    again = (pixels.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0))
    assert np.array_equal(again.view(np.uint32), full.view(np.uint32)), "u8 round trip is not exact"
    olib = common.oracle_lib()
    mine = common.oracle_load_resized(olib, pixels, W, H)
    assert np.array_equal(mine.view(np.uint32), sized.view(np.uint32)), "oracle front end != reference on dog.jpg"
    keep = {}
    cfg, wts = common.model_files(NAME, W, H)
    for q, tag in ((0, "fp32"), (1, "int8")):
        ref = refbind.RefNetwork(cfg, wts, 1, q)
        ref.predict(sized[None])
        sums = np.zeros((ref.n, 2), np.float64)
        for i in range(ref.n):
            o = ref.layer_output(i).astype(np.float64)
            sums[i] = (o.sum(), np.abs(o).sum())
        dets = ref.get_detections(0, sw, sh, 0.24, nms=0.4, relative=1)
        keep[tag + "_layer_sums"] = sums
        keep[tag + "_dets"] = dets
        if q == 0:
            # the two [yolo] head tensors of the FP32 path, element by element (0.86 MB): the GPU test compares them at
            # north_star's 1e-4 without oracle/_ref on the box (VERDICT round 4, weak 1 i)
            heads = [i for i in range(ref.n) if ref.layer_info(i)["type"] == common.YOLO]
            keep["fp32_head_layers"] = np.array(heads)
            for i in heads:
                keep["fp32_head_%d" % i] = ref.layer_output(i).astype(np.float32)
        # random weights leave few boxes above .24: a second, dense set at a low threshold
        lo = ref.get_detections(0, sw, sh, LOW_THRESH, nms=0.4, relative=1)
        keep[tag + "_dets_low"] = lo
        print(tag, len(dets), "detections,", len(lo), "at thresh", LOW_THRESH)
    out = os.path.join(HERE, "dog", "dog_yolov3-tiny_416.npz")
    np.savez_compressed(out, pixels=pixels, src_wh=np.array([sw, sh]),
                        sized_sha256=np.array(hashlib.sha256(sized.tobytes()).hexdigest()),
                        low_thresh=np.array(LOW_THRESH),
                        weights_sha256=np.array(hashlib.sha256(open(wts, "rb").read()).hexdigest()), **keep)
    print(out, os.path.getsize(out), "bytes")


def main_xnor():
    lib = refbind._bind(refbind.GOLD)
    sized, sw, sh = load_resized(lib, DOG, W, H)
    name = "tiny-yolo-xnor"
    cfg, wts = common.model_files(name, W, H)
    ref = refbind.RefNetwork(cfg, wts, 1, 0)
    ref.predict(sized[None])
    keep = {}
    sums = np.zeros((ref.n, 2), np.float64)
    xnor_layers = []
    for i in range(ref.n):
        o = ref.layer_output(i)
        sums[i] = (o.astype(np.float64).sum(), np.abs(o.astype(np.float64)).sum())
        li = ref.layer_info(i)
        if li["type"] == common.CONV and li["xnor"] and li["size"] == 3 and li["stride"] == 1 and li["pad"] == 1:
            src = ref.layer_output(i - 1)
            assert src.size == li["c"] * li["h"] * li["w"]
            keep["in_bits_%d" % i] = np.packbits(src > 0)              # bit = (x > 0), CHW order
            keep["out_sha256_%d" % i] = np.array(hashlib.sha256(o.tobytes()).hexdigest())
            xnor_layers.append(i)
    keep["xnor_layers"] = np.array(xnor_layers)
    keep["layer_sums"] = sums
    for key, thresh in (("dets", 0.24), ("dets_low", LOW_THRESH)):
        keep[key] = ref.get_detections(0, sw, sh, thresh, nms=0.4, relative=1)
        print(key, len(keep[key]), "detections at thresh", thresh)
    out = os.path.join(HERE, "dog", "dog_tiny-yolo-xnor_416.npz")
    np.savez_compressed(out, src_wh=np.array([sw, sh]),
                        sized_sha256=np.array(hashlib.sha256(sized.tobytes()).hexdigest()),
                        low_thresh=np.array(LOW_THRESH),
                        weights_sha256=np.array(hashlib.sha256(open(wts, "rb").read()).hexdigest()), **keep)
    print(out, os.path.getsize(out), "bytes; xnor layers", xnor_layers)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("tiny", "all"):
        main()
    if what in ("xnor", "all"):
        main_xnor()
