"""Parity on the configurations the metric is quoted on (BASELINE.json configs 1, 3, 4):

  * yolov3 608x608 batch 64 FP32 with fusion on (= the benched setup): images 0, 31, 63 equal, bit for
    bit on every materialised tensor and on the detection rows, a batch-1 run of the same image; the
    batch-1 fused run equals the batch-1 unfused run, which tests/test_gpu_parity.py pins to the
    reference library at this size.  At batch 64 tile counts, 32-bit buffer offsets (layer 0 writes
    3.03 GB > 2^31) and the fused-shortcut / Winograd dispatch differ from batch 1.
  * the same for `-quantized` at 608, plus every layer of a 608x608 batch-1 INT8 run checked
    teacher-forced (each layer against the oracle applied to the GPU's own input of that layer):
    accumulators and outputs bit-exact.  The INT8 convolutions use oracle/fast_oracle.c (exact
    integer reordering of oracle_conv_int8, pinned to it in tests/test_oracle_pin.py).
  * dog.jpg (config 1): the reference's decoded photo -> GPU front end -> yolov3-tiny 416 against the
    fixture the reference CPU path produced (tests/golden/make_golden_dog.py), and through the
    reference's own host code + network_predict_hip when oracle/_ref is present.
  * yolov3-tiny 416 batch 32 (config 2) and tiny-yolo-xnor 416 batch 128 (config 5), as bench.py's side legs run
    them, against batch-1 runs, bit for bit; the kernel instances the bench line names are asserted.
  * dog.jpg through the XNOR network (config 5's topology): detections against the reference CPU path's, every layer
    teacher-forced against the oracle, and each XNOR convolution on the sign bits the REFERENCE fed it reproduces
    the sha256 of what the reference got out (bit-exact against the reference itself, no oracle in between).
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import common
from common import Network, fp, fp32_close, refbind

pytestmark = pytest.mark.gpu

_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)
IMAGES = (0, 31, 63)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _dominant_kernel(net, quantized):
    """the kernel instance bench.py's roofline block calls dominant (most algorithmic FLOPs; conv_xnor's template
    instances count as one kernel there)"""
    flops = {}
    for i, li in enumerate(net.layers()):
        if li["type"] == common.CONV:
            k = net.layer_kernel(i).replace(",pool+", "").replace(",pool", "")
            if quantized and not k.startswith("conv_i8"):
                continue
            if k.startswith("conv_xnor"):
                k = "conv_xnor"
            flops[k] = flops.get(k, 0.0) + 2.0 * li["n"] * li["size"] ** 2 * li["c"] * li["out_h"] * li["out_w"]
    return max(flops, key=flops.get)


def _big_batch_equals_batch1(name, size, B, quantized, images, min_checked, dets_range=(50, 4096), expect_kernels=None):
    """the benched configuration (fusion on, batch B) against batch-1 runs of the same images, bit for bit on every
    materialised tensor and on the detection rows; then fused == unfused at batch 1 (the unfused run is what
    tests/test_gpu_parity.py / test_gpu_int8_xnor.py pin to the oracle and the reference library)"""
    cfg, wts = common.model_files(name, size, size)
    x = common.seeded_input(B, 3, size, size)
    big = Network.load(cfg, wts, B, quantized, device=0, fuse=True)
    # ONE pass over the whole batch, as bench.py's timed step runs it (yl_network_predict would pipeline two sub-batches, whose
    # grids -- and with them the tile heuristics the kernel names below are asserted on -- are those of B / 2 images;
    # tests/test_gpu_dropin.py::test_pipelined_predict_equals_one_pass pins the pipelined form to the one-pass bits)
    old_split = os.environ.get("YL_PREDICT_SPLIT")
    os.environ["YL_PREDICT_SPLIT"] = "1"
    try:
        big.predict(x)
    finally:
        if old_split is None:
            os.environ.pop("YL_PREDICT_SPLIT", None)
        else:
            os.environ["YL_PREDICT_SPLIT"] = old_split
    # the kernel instance bench.py's roofline block will call dominant at this configuration must have its committed
    # PMC traffic entry (profiles/pmc_traffic.json): a renamed or re-tiled kernel fails HERE, not as `traffic: null`
    import json
    dominant = _dominant_kernel(big, quantized)
    key = dominant if (name == "yolov3" and size == 608) else "%s@%s-%d" % (dominant, name, size)
    with open(os.path.join(common.ROOT, "profiles", "pmc_traffic.json")) as f:
        assert key in json.load(f), "no PMC traffic entry for the dominant kernel %r" % key
    for i, want in (expect_kernels or {}).items():
        assert big.layer_kernel(i) == want, "layer %d runs %r at batch %d, the bench line is quoted on %r" % (
            i, big.layer_kernel(i), B, want)
    one = Network.load(cfg, wts, 1, quantized, device=0, fuse=True)
    plain = Network.load(cfg, wts, 1, quantized, device=0, fuse=False)
    n_checked = 0
    for b in images:
        one.predict(x[b:b + 1])
        for i in range(one.n):
            if not one.layer_materialised(i):
                assert not big.layer_materialised(i)
                continue
            a = big.layer_output_image(i, b)
            r = one.layer_output(i)
            assert np.array_equal(_bits(a), _bits(r)), "image %d layer %d %r: batch-%d != batch-1" % (b, i, one.layer_info(i), B)
            n_checked += 1
        rows_big = big.get_boxes(b, size, size, 0.24, nms=0.4)
        rows1 = one.get_boxes(0, size, size, 0.24, nms=0.4)
        assert rows_big.shape == rows1.shape and np.array_equal(_bits(rows_big), _bits(rows1)), "image %d detection rows" % b
        # the untrained head passes thousands of cells at .24 (a dense set): it must stay below the device
        # path's capacity for the order to be defined
        assert dets_range[0] < len(rows1) < dets_range[1], len(rows1)
    # fused == unfused at batch 1 (the unfused run is what is pinned to the reference library)
    plain.predict(x[images[-1]:images[-1] + 1])
    n_fused = 0
    for i in range(one.n):
        if one.layer_materialised(i):
            assert np.array_equal(_bits(plain.layer_output(i)), _bits(one.layer_output(i))), "fused != unfused, layer %d" % i
        else:
            n_fused += 1
    assert n_checked >= min_checked, n_checked
    big.close(); one.close(); plain.close()
    return n_fused


def test_yolov3_608_batch64_fp32_fused_equals_batch1():
    _big_batch_equals_batch1("yolov3", 608, 64, 0, IMAGES, 3 * 60 + 1)


def test_yolov3_608_batch64_image_against_the_reference_library_directly():
    """The benched configuration (batch 64, fusion on) compared with the reference CPU path DIRECTLY, not through the
    batch-1 chain: image 17 of the batch -- one that the bit-equality test above does not
    read back -- every materialised tensor against network_predict_cpu of the unmodified reference on that image alone
    (batch B == B independent images), at the FP32 contract's fp32_close and the strict
    per-layer relative bound test_full_size_vs_reference_library uses."""
    common.require_ref()
    size, B, b = 608, 64, 17
    cfg, wts = common.model_files("yolov3", size, size)
    x = common.seeded_input(B, 3, size, size)
    big = Network.load(cfg, wts, B, 0, device=0, fuse=True)
    big.predict(x)
    ref = refbind.RefNetwork(cfg, wts, 1, 0)
    ref.predict(x[b:b + 1])
    n = 0
    worst_ratio = worst_strict = 0.0
    for i in range(big.n):
        if not big.layer_materialised(i):
            continue
        got = big.layer_output_image(i, b)
        want = ref.layer_output(i)
        ok, ratio, worst = fp32_close(got, want)
        assert ok, "batch-64 image %d layer %d %r: err/allowed %.3g" % (b, i, big.layer_info(i), ratio)
        strict = common.strict_max_rel(got, want)
        assert strict <= 2e-2, "batch-64 image %d layer %d: strict max relative error %.3g" % (b, i, strict)
        worst_ratio, worst_strict = max(worst_ratio, ratio), max(worst_strict, strict)
        n += 1
    assert n >= 60
    r = ref.get_detections(0, size, size, 0.24, nms=0.4)
    g = big.get_boxes(b, size, size, 0.24, nms=0.4, relative=1)
    assert abs(len(r) - len(g)) <= 2, "%d vs %d detections" % (len(r), len(g))
    print("yolov3 608 batch 64 image %d vs the reference: %d tensors, worst fp32_close ratio %.3g, strict max-rel %.3g, %d / %d detections"
          % (b, n, worst_ratio, worst_strict, len(g), len(r)))
    big.close()


def test_yolov3_608_batch64_int8_fused_equals_batch1():
    _big_batch_equals_batch1("yolov3", 608, 64, 1, IMAGES, 3 * 40)      # (40 tensors per image: the two upsampled ones in front of the multi-input routes are never written)


def test_yolov3_tiny_416_batch32_fp32_fused_equals_batch1():
    """BASELINE config 2 as bench.py's side leg runs it (yolov3-tiny 416, batch 32, fusion on): images 0 / 15 / 31."""
    _big_batch_equals_batch1("yolov3-tiny", 416, 32, 0, (0, 15, 31), 3 * 12, dets_range=(10, 4096))


def test_tiny_yolo_xnor_416_batch128_fused_equals_batch1():
    """BASELINE config 5 as bench.py's side leg runs it (tiny-yolo-obj_xnor 416, batch 128, sign-domain fusion on):
    images 0 / 64 / 127.  conv_xnor's filter tile is chosen by GRID depth (conv_xnor.hip, launch_conv_xnor): at batch
    128 the 208 x 208 and 104 x 104 layers run 64-filter workgroups where the layer has 64 filters and the
    13 x 13 layers 32-filter ones; at batch 1 everything runs 32-filter tiles -- so this is also 64-filter == 32-filter
    tiles on the whole network.  XNOR arithmetic is integer: bit for bit, no tolerance (north_star)."""
    cfg, _ = common.model_files("tiny-yolo-xnor", 416, 416)
    probe = Network.from_cfg(cfg, 1, 0)
    xnor_layers = [i for i, li in enumerate(probe.layers()) if li["type"] == common.CONV and li["xnor"]]
    infos = probe.layers()
    probe.close()
    assert len(xnor_layers) == 7
    expect = {}
    for i in xnor_layers:
        li = infos[i]
        wg64 = -(-128 * li["h"] * li["w"] // 256) * -(-li["n"] // 64)
        ft = 32 if (li["n"] < 64 or wg64 < 16 * 256) else 64
        expect[i] = "conv_xnor<ft%d,%s,thr>" % (ft, "w32" if li["c"] <= 32 else "w64")
    # the layer in front of the region head's linear conv feeds an FP32 conv: float epilogue, no thresholds
    expect[xnor_layers[-1]] = expect[xnor_layers[-1]].replace(",thr", "")
    assert any("ft64" in v for v in expect.values()) and any("ft32" in v for v in expect.values())
    n_fused = _big_batch_equals_batch1("tiny-yolo-xnor", 416, 128, 0, (0, 64, 127), 3 * 3, dets_range=(0, 4096),
                                       expect_kernels=expect)
    assert n_fused >= 8          # the sign domain really was on: FP32 tensors between the bit layers never existed


def test_yolov3_608_int8_every_layer_teacher_forced(olib):
    name, size = "yolov3", 608
    fast = common.oracle_fast_lib()
    cfg, wts = common.model_files(name, size, size)
    net = Network.load(cfg, wts, 1, 1, device=0, debug=True)
    x = common.seeded_input(1, 3, size, size, seed=77)
    net.predict(x)
    infos = net.layers()
    from yolo2_light_amd import zoo
    route_inputs = {}
    for i, (typ, o) in enumerate(zoo.parse_sections(open(cfg).read())[1:]):
        if typ == "route":
            ids = [int(v) for v in o["layers"].split(",")]
            route_inputs[i] = [j + i if j < 0 else j for j in ids]
    outs = {}

    def out(i):
        if i not in outs:
            outs[i] = net.layer_output(i)
        return outs[i]

    n_int8 = 0
    for i, li in enumerate(infos):
        cur = x.reshape(-1) if i == 0 else out(i - 1)
        ref = np.zeros(li["outputs"], np.float32)
        t = li["type"]
        exact = True
        if t == common.CONV:
            b_ = net.layer_biases(i)
            if li["conv_mode"] == common.CONV_INT8:
                wq = net.layer_weights_int8(i)
                im, wm = net.layer_quant_multipliers(i)
                racc = np.zeros(ref.size, np.int32)
                fast.oracle_conv_int8_fast(fp(cur), wq.ctypes.data_as(_i8p), fp(b_), fp(ref), racc.ctypes.data_as(_i32p),
                                           1, li["c"], li["h"], li["w"], li["n"], li["size"], li["stride"], li["pad"],
                                           li["activation"], im, wm)
                assert np.array_equal(net.layer_int8_acc(i), racc), "layer %d int8 accumulators" % i
                n_int8 += 1
            else:
                olib.oracle_conv_f32(fp(cur), fp(net.layer_weights(i)), fp(b_), fp(ref), 1, li["c"], li["h"], li["w"],
                                     li["n"], li["size"], li["stride"], li["pad"], li["activation"])
                exact = False
        elif t == common.ROUTE:
            ref = np.concatenate([out(j) for j in route_inputs[i]])
        elif t == common.SHORTCUT:
            olib.oracle_shortcut(fp(cur), fp(out(li["index"])), fp(ref), 1, li["w"], li["h"], li["c"],
                                 li["out_w"], li["out_h"], li["out_c"], li["activation"])
        elif t == common.UPSAMPLE:
            olib.oracle_upsample(fp(cur), fp(ref), 1, li["c"], li["h"], li["w"], li["stride"], 1.0)
        elif t == common.YOLO:
            olib.oracle_yolo(fp(cur), fp(ref), 1, li["n"], li["classes"], li["w"] * li["h"])
            exact = False
        else:
            raise AssertionError("unexpected layer type %d" % t)
        got = out(i)
        if exact:
            assert np.array_equal(_bits(got), _bits(ref)), "layer %d %r not bit-exact" % (i, li)
        else:
            ok, ratio, worst = fp32_close(got, ref)
            assert ok, "layer %d %r: err/allowed %.3g" % (i, li, ratio)
    assert n_int8 == 71
    net.close()


# ---------------------------------------------------------------------------- dog.jpg (config 1)
DOG = os.path.join(common.GOLDEN_DIR, "dog", "dog_yolov3-tiny_416.npz")


def _match_rows(r, g):
    """fraction of reference rows r that have a row of g with the same box (1e-4 rel)"""
    if not len(r) or not len(g):
        return 1.0 if len(r) == len(g) else 0.0, None
    with np.errstate(invalid="ignore", over="ignore"):
        dist = (np.abs(r[:, None, :4] - g[None, :, :4]) / (1e-5 + 1e-4 * np.abs(r[:, None, :4]))).max(axis=2)
    dist = np.nan_to_num(dist, nan=0.0)
    j = dist.argmin(axis=1)
    ok = dist[np.arange(len(r)), j] < 1.0
    return ok.mean(), (ok, j)


def test_dog_jpg_fp32_against_the_reference_cpu_fixture():
    z = np.load(DOG)
    pixels = z["pixels"]
    sw, sh = (int(v) for v in z["src_wh"])
    name, W, H = "yolov3-tiny", 416, 416
    cfg, wts = common.model_files(name, W, H)
    assert hashlib.sha256(open(wts, "rb").read()).hexdigest() == str(z["weights_sha256"]), "synthetic weights changed"
    net = Network.load(cfg, wts, 1, 0, device=0)
    net.set_input_u8(0, pixels)                               # GPU /255 + resize_image of the decoded photo
    sized = net.input_download()
    assert hashlib.sha256(sized.tobytes()).hexdigest() == str(z["sized_sha256"]), "front end differs from the reference's sized.data"
    net.forward_staged()
    net.synchronize()
    sums = z["fp32_layer_sums"]
    for i in range(net.n):
        o = net.layer_output(i).astype(np.float64)
        # sum of N terms each within 1e-4 relative: compare against the abs-sum scale
        assert abs(o.sum() - sums[i, 0]) <= 1e-4 * sums[i, 1] + 1e-6, "layer %d sum" % i
        assert abs(np.abs(o).sum() - sums[i, 1]) <= 1e-4 * sums[i, 1] + 1e-6, "layer %d abs-sum" % i
    # the [yolo] head tensors element by element at north_star's 1e-4 (|err| <= 1e-4 |ref| + 1e-4 RMS, common.fp32_close)
    assert len(z["fp32_head_layers"]) == 2
    for i in z["fp32_head_layers"]:
        ok, ratio, worst = fp32_close(net.layer_output(int(i)), z["fp32_head_%d" % int(i)].reshape(-1))
        assert ok, "head layer %d: err/allowed %.3g at element %d" % (int(i), ratio, worst)
    for key, thresh in (("fp32_dets", 0.24), ("fp32_dets_low", float(z["low_thresh"]))):
        r = z[key]
        g = net.get_boxes(0, sw, sh, thresh, nms=0.4)
        assert abs(len(r) - len(g)) <= max(2, len(r) // 50), (key, len(r), len(g))
        frac, m = _match_rows(r, g)
        assert frac > 0.98, (key, frac)
        if m is not None:
            ok, j = m
            np.testing.assert_allclose(g[j[ok]][:, 4], r[ok][:, 4], rtol=1e-4, atol=1e-5)
    assert len(z["fp32_dets_low"]) > 100
    net.close()


@pytest.mark.parametrize("quantized", [0, 1])
def test_dog_jpg_reference_host_code_cpu_vs_hip(quantized):
    """src/main.c:187-229 on the photo: load_image + resize_image (fixture pixels through the pinned
    oracle front end = the reference's sized.data), then network_predict_cpu / _quantized vs
    network_predict_hip on the SAME network object, then the reference's own get_network_boxes + do_nms_sort."""
    common.require_ref(hip=True)
    z = np.load(DOG)
    sw, sh = (int(v) for v in z["src_wh"])
    name, W, H = "yolov3-tiny", 416, 416
    cfg, wts = common.model_files(name, W, H)
    sized = common.oracle_load_resized(common.oracle_lib(), z["pixels"], W, H)
    assert hashlib.sha256(sized.tobytes()).hexdigest() == str(z["sized_sha256"])
    ref = refbind.RefNetwork(cfg, wts, 1, quantized, hip=True)
    x = sized[None]
    ref.predict(x)
    heads = [i for i in range(ref.n) if ref.layer_info(i)["type"] == common.YOLO]
    cpu_heads = {i: ref.layer_output(i) for i in heads}
    thresh = float(z["low_thresh"])
    cpu_dets = ref.get_detections(0, sw, sh, thresh, nms=0.4)
    key = "int8_dets_low" if quantized else "fp32_dets_low"
    assert np.array_equal(_bits(cpu_dets), _bits(z[key])), "the reference no longer reproduces its own fixture"
    ref.predict_hip(x)
    for i in heads:
        got, want = ref.layer_output(i), cpu_heads[i]
        if quantized:
            # int8 codes are step functions of the (1e-6-different) FP32 first layer: statistical agreement,
            # layer exactness is established teacher-forced
            g = got.astype(np.float64); r = want.astype(np.float64)
            assert np.sqrt(np.mean((g - r) ** 2)) / np.sqrt(np.mean(r * r)) < 0.05
        else:
            ok, ratio, _ = fp32_close(got, want)
            assert ok, "head %d: err/allowed %.3g" % (i, ratio)
    hip_dets = ref.get_detections(0, sw, sh, thresh, nms=0.4)
    if not quantized:
        assert abs(len(hip_dets) - len(cpu_dets)) <= max(2, len(cpu_dets) // 50)
        frac, _ = _match_rows(cpu_dets, hip_dets)
        assert frac > 0.98
    ref.lib.ref_free_hip()


# ---------------------------------------------------------------------------- dog.jpg through the XNOR network
DOG_XNOR = os.path.join(common.GOLDEN_DIR, "dog", "dog_tiny-yolo-xnor_416.npz")


def _dog_sized_on_gpu(net):
    z = np.load(DOG)
    net.set_input_u8(0, z["pixels"])
    sized = net.input_download()
    assert hashlib.sha256(sized.tobytes()).hexdigest() == str(z["sized_sha256"])
    return sized


def test_dog_jpg_xnor_network_against_the_reference_cpu_fixture():
    """src/main.c:187-229 with bin/tiny-yolo-obj_xnor.cfg's topology: photo -> GPU front end -> FP32 first layer ->
    7 XNOR convolutions -> region head, against what the reference CPU path produced for the same photo and weights.
    End to end the comparison is tolerant (a last-bit difference of the FP32 first layer can flip a sign bit); the
    bit-exact statements are the two tests below."""
    z = np.load(DOG_XNOR)
    sw, sh = (int(v) for v in z["src_wh"])
    name, W, H = "tiny-yolo-xnor", 416, 416
    cfg, wts = common.model_files(name, W, H)
    assert hashlib.sha256(open(wts, "rb").read()).hexdigest() == str(z["weights_sha256"]), "synthetic weights changed"
    net = Network.load(cfg, wts, 1, 0, device=0)
    _dog_sized_on_gpu(net)
    net.forward_staged()
    net.synchronize()
    sums = z["layer_sums"]
    worst = 0.0
    for i in range(net.n):
        o = net.layer_output(i).astype(np.float64)
        worst = max(worst, abs(o.sum() - sums[i, 0]) / (sums[i, 1] + 1e-6), abs(np.abs(o).sum() - sums[i, 1]) / (sums[i, 1] + 1e-6))
        assert abs(o.sum() - sums[i, 0]) <= 1e-3 * sums[i, 1] + 1e-6, "layer %d sum" % i
        assert abs(np.abs(o).sum() - sums[i, 1]) <= 1e-3 * sums[i, 1] + 1e-6, "layer %d abs-sum" % i
    print("dog.jpg xnor: worst layer-sum deviation / abs-sum = %.3g" % worst)
    rows = {}
    for key, thresh in (("dets", 0.24), ("dets_low", float(z["low_thresh"]))):
        r = z[key]
        g = net.get_boxes(0, sw, sh, thresh, nms=0.4)
        rows[key] = g
        assert abs(len(r) - len(g)) <= max(2, len(r) // 50), (key, len(r), len(g))
        frac, m = _match_rows(r, g)
        assert frac > 0.98, (key, frac)
        if m is not None:
            ok, j = m
            np.testing.assert_allclose(g[j[ok]][:, 4], r[ok][:, 4], rtol=1e-3, atol=1e-5)
    assert len(z["dets"]) > 100
    net.close()
    # the benched form (sign-domain fusion on) returns the same rows, bit for bit
    fused = Network.load(cfg, wts, 1, 0, device=0, fuse=True)
    _dog_sized_on_gpu(fused)
    fused.forward_staged()
    fused.synchronize()
    g = fused.get_boxes(0, sw, sh, 0.24, nms=0.4)
    assert g.shape == rows["dets"].shape and np.array_equal(_bits(g), _bits(rows["dets"]))
    fused.close()


def test_dog_jpg_xnor_every_layer_teacher_forced(olib):
    """every layer of the XNOR network on the photo against the oracle applied to the GPU's own input of that layer:
    match counts and outputs of the 7 XNOR convolutions, max-pools and the region layout bit-exact"""
    from test_gpu_int8_xnor import _teacher_forced
    z = np.load(DOG)
    sized = common.oracle_load_resized(olib, z["pixels"], 416, 416)
    assert hashlib.sha256(sized.tobytes()).hexdigest() == str(z["sized_sha256"])
    stats = _teacher_forced(olib, "tiny-yolo-xnor", 416, 416, 1, 0, x=sized[None])
    assert stats["exact"] >= 7 + 6, stats


def test_dog_jpg_xnor_layers_bit_exact_against_the_reference_fixture():
    """Each XNOR convolution of the network, alone, on the sign bits the REFERENCE CPU path fed it for dog.jpg
    (fixture, tests/golden/make_golden_dog.py xnor): the output tensor's sha256 equals the reference's.  The bit path
    reads nothing but signs (src/yolov2_forward_network.c:116-203), so +-1 stands for the reference's tensor."""
    import descs as D
    z = np.load(DOG_XNOR)
    name, W, H = "tiny-yolo-xnor", 416, 416
    cfg, wts = common.model_files(name, W, H)
    assert hashlib.sha256(open(wts, "rb").read()).hexdigest() == str(z["weights_sha256"]), "synthetic weights changed"
    model = Network.load(cfg, wts, 1, 0)            # host side only: fused weights, biases, mean_arr
    layers = [int(i) for i in z["xnor_layers"]]
    assert len(layers) == 7
    for i in layers:
        li = model.layer_info(i)
        n_in = li["c"] * li["h"] * li["w"]
        bits = np.unpackbits(z["in_bits_%d" % i])[:n_in].astype(bool)
        x = np.where(bits, np.float32(1.0), np.float32(-1.0)).reshape(1, li["c"], li["h"], li["w"])
        d = D.conv(1, li["w"], li["h"], li["c"], li["n"], 3, 1, 1, li["activation"], model.layer_weights(i),
                   model.layer_biases(i), xnor=1, mean_arr=model.layer_mean_arr(i))
        for variant in (30, 30 | 512):            # filter tile by grid depth (32 at batch 1) / 64-filter workgroups
            net = Network.from_desc([d], 1, li["w"], li["h"], li["c"], 0)
            net.set_variant(variant)
            net.to_device(0)
            got = net.predict(x)
            assert hashlib.sha256(np.ascontiguousarray(got, dtype=np.float32).tobytes()).hexdigest() == str(z["out_sha256_%d" % i]), \
                "XNOR layer %d (%dx%d, %d -> %d) differs from the reference CPU path on dog.jpg (variant %d)" % (
                    i, li["w"], li["h"], li["c"], li["n"], variant)
            net.close()
    model.close()
