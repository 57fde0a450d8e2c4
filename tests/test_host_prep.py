"""Host-side model build: our cfg parser + prep passes (BN fuse, binary means,
INT8 weight quantisation + multipliers) against the reference's own
(oracle/_ref) and against the oracle restatement.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

import common
from common import Network, fp, refbind

GEOM = ("type", "w", "h", "c", "n", "size", "stride", "pad", "out_w", "out_h", "out_c", "outputs", "inputs")

MODELS = [("yolov3-tiny", 416, 416), ("yolov3", 608, 608), ("tiny-yolo-xnor", 416, 416), ("yolov3", 96, 160)]


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,width,height", MODELS)
@pytest.mark.parametrize("quantized", [0, 1])
def test_parser_and_prep_match_reference(name, width, height, quantized):
    cfg, wts = common.model_files(name, width, height)
    ref = refbind.RefNetwork(cfg, wts, 2, quantized)
    net = Network.load(cfg, wts, 2, quantized)
    assert ref.n == net.n
    convs = 0
    for i in range(net.n):
        a, b = net.layer_info(i), ref.layer_info(i)
        for k in GEOM:
            if a["type"] in (common.ROUTE, common.REGION) and k in ("w", "h", "c", "n", "out_w", "out_h", "out_c", "size", "stride"):
                continue     # fields the reference leaves unset for these layers
            if a["type"] in (common.SHORTCUT, common.YOLO, common.UPSAMPLE, common.MAXPOOL) and k in ("size", "stride", "pad", "n"):
                if a["type"] == common.MAXPOOL or (a["type"] == common.UPSAMPLE and k == "stride") or (a["type"] == common.YOLO and k == "n"):
                    pass
                else:
                    continue
            assert a[k] == b[k], (i, k, a[k], b[k])
        if a["type"] == common.CONV:
            convs += 1
            assert a["activation"] == b["activation"] and a["xnor"] == b["xnor"]
            assert np.array_equal(net.layer_weights(i).view(np.uint32), ref.layer_weights(i).view(np.uint32)), i
            assert np.array_equal(net.layer_biases(i).view(np.uint32), ref.layer_biases(i).view(np.uint32)), i
            if a["xnor"]:
                assert np.array_equal(net.layer_mean_arr(i).view(np.uint32), ref.layer_mean_arr(i).view(np.uint32)), i
            if quantized:
                assert np.array_equal(net.layer_weights_int8(i), ref.layer_weights_int8(i)), i
                assert net.layer_quant_multipliers(i) == ref.layer_quant_multipliers(i), i
        if a["type"] == common.SHORTCUT:
            assert a["index"] == b["index"]
        if a["type"] in (common.YOLO, common.REGION):
            assert a["classes"] == b["classes"] and a["n"] == b["n"]
    assert convs > 0


def test_prep_matches_oracle_restatement(olib):
    """The same passes against oracle/yolo2_oracle.c (runs even without oracle/_ref)."""
    rng = np.random.default_rng(4)
    n, fs = 7, 45
    w = rng.normal(0, 0.3, n * fs).astype(np.float32)
    b = rng.normal(0, 0.1, n).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, n).astype(np.float32)
    mu = rng.normal(0, 0.1, n).astype(np.float32)
    var = rng.uniform(0.5, 1.5, n).astype(np.float32)
    import descs as D
    d = D.conv(1, 5, 5, 5, n, 3, 1, 1, D.LEAKY, w, b, xnor=1)
    d.batch_normalize = 1
    d.scales = fp(sc); d.rolling_mean = fp(mu); d.rolling_variance = fp(var)
    l0 = D.maxpool(1, 5, 5, 5, 1, 1, pad=0)
    net = Network.from_desc([l0, d], 1, 5, 5, 5, quantized=1)
    net.fuse_conv_batchnorm(); net.calculate_binary_weights(); net.quantize()
    w2, b2 = w.copy(), b.copy()
    olib.oracle_fuse_bn(fp(w2), fp(b2), fp(sc), fp(mu), fp(var), n, fs)
    assert np.array_equal(net.layer_weights(1).view(np.uint32), w2.view(np.uint32))
    assert np.array_equal(net.layer_biases(1).view(np.uint32), b2.view(np.uint32))
    mean = np.zeros(n, np.float32)
    olib.oracle_binary_mean(fp(w2), n, fs, fp(mean))
    assert np.array_equal(net.layer_mean_arr(1).view(np.uint32), mean.view(np.uint32))
    wq = np.zeros(n * fs, np.int8)
    wm = olib.oracle_quantize_weights(fp(w2), n * fs, wq.ctypes.data_as(C.POINTER(C.c_int8)))
    assert np.array_equal(net.layer_weights_int8(1), wq)
    assert net.layer_quant_multipliers(1)[1] == wm
    assert net.layer_quant_multipliers(1)[0] == 40.0          # no calibration list -> 40


def test_cfg_errors_are_reported(tmp_path):
    from yolo2_light_amd import YoloHipError
    p = tmp_path / "bad.cfg"
    p.write_text("[net]\nwidth=32\nheight=32\nchannels=3\n[convolutional]\nfilters=8\nsize=3\nstride=1\npad=1\nactivation=leaky\n[yolo]\nmask=0\nanchors=1,2\nclasses=80\nnum=1\n")
    with pytest.raises(YoloHipError):
        Network.from_cfg(str(p), 1, 0)           # filters != n*(classes+5)
    with pytest.raises(YoloHipError):
        Network.from_cfg(str(tmp_path / "missing.cfg"), 1, 0)
    p2 = tmp_path / "short.cfg"
    p2.write_text("[net]\nwidth=32\nheight=32\nchannels=3\n[convolutional]\nfilters=8\nsize=3\nstride=1\npad=1\nactivation=leaky\n")
    net = Network.from_cfg(str(p2), 1, 0)
    wfile = tmp_path / "short.weights"
    wfile.write_bytes(b"\0" * 40)
    with pytest.raises(YoloHipError):
        net.load_weights(str(wfile))             # truncated file is refused, not half-loaded


@pytest.mark.parametrize("C,M,tiling", [(16, 33, 32), (24, 64, 32), (64, 70, 32), (32, 130, 32)])
def test_winograd_weight_packing(C, M, tiling):
    """U = G g G^T (double, rounded once) lands where the K1w kernel reads it (conv_f32_wino32.hip);
    filters beyond M are zero."""
    import ctypes as Cc
    from yolo2_light_amd._lib import lib
    rng = np.random.default_rng(C * 100 + M)
    w = rng.standard_normal((M, C, 3, 3)).astype(np.float32)
    fp = Cc.POINTER(Cc.c_float)
    need = lib.yl_debug_wino_pack(w.ctypes.data_as(fp), C, M, tiling, None, 0)
    assert lib.yl_debug_wino_pack(w.ctypes.data_as(fp), C, M, 16, None, 0) < 0       # round 3's other packings are gone
    BM, BK = 32, 4    # filters x 4-channel panels per workgroup
    tiles_m = (M + BM - 1) // BM
    assert need == tiles_m * (C // BK) * 16 * BK * BM
    dst = np.full(need, np.nan, dtype=np.float32)
    assert lib.yl_debug_wino_pack(w.ctypes.data_as(fp), C, M, tiling, dst.ctypes.data_as(fp), need) == need
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    g = w.astype(np.float64)
    # same evaluation order as the C code: t = G g (left to right), u = t G^T (left to right)
    t = np.zeros((M, C, 4, 3))
    for i in range(4):
        t[:, :, i, :] = G[i, 0] * g[:, :, 0, :] + G[i, 1] * g[:, :, 1, :] + G[i, 2] * g[:, :, 2, :]
    u = np.zeros((M, C, 4, 4))
    for j in range(4):
        u[:, :, :, j] = t[:, :, :, 0] * G[j, 0] + t[:, :, :, 1] * G[j, 1] + t[:, :, :, 2] * G[j, 2]
    u = u.astype(np.float32)
    KK = BK // 2
    p = dst.reshape(tiles_m, C // BK, 16, 2, BM, KK)      # [tile_m][panel][xi][half][m][kk]
    for tm in range(tiles_m):
        for ml in range(BM):
            m = tm * BM + ml
            for c in range(C):
                kb, kl = divmod(c, BK)
                got = p[tm, kb, :, kl & 1, ml, kl >> 1]
                want = u[m, c].reshape(16) if m < M else np.zeros(16, np.float32)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tm, ml, c)


@pytest.mark.parametrize("C,M,size", [(16, 33, 3), (32, 128, 1), (48, 70, 3), (64, 255, 1), (32, 20, 5)])
def test_x3_weight_split_is_exact(C, M, size):
    """conv_f32_x3.hip's weights: every FP32 weight as three bf16 numbers.  Their sum is the weight EXACTLY (the kernel's
    products are then FP32-exact up to the three dropped cross terms), each piece is the round-to-nearest-even bf16 of what
    the pieces before it left, and the units sit where the kernel's lanes read them; filters beyond M are zero."""
    import ctypes as Cc
    from yolo2_light_amd._lib import lib
    rng = np.random.default_rng(C * 1000 + M + size)
    # nine octaves of magnitude, a few exact zeros and exact bf16 values
    w = (rng.standard_normal((M, C, size, size)) * np.exp(rng.uniform(-4, 2, (M, C, size, size)))).astype(np.float32)
    w[rng.random(w.shape) < 0.02] = 0.0
    w[0, 0] = np.float32(1.5)
    fp = Cc.POINTER(Cc.c_float)
    assert lib.yl_debug_x3_pack(w.ctypes.data_as(fp), C + 1, M, size, None, 0) < 0            # C % 16 != 0: not this kernel's layer
    need = lib.yl_debug_x3_pack(w.ctypes.data_as(fp), C, M, size, None, 0)
    taps = size * size
    mpad = (M + 127) // 128 * 128
    assert need == (C // 16) * taps * 6 * mpad * 16
    raw = np.full(need // 2, 0x7fc0, dtype=np.uint16)
    assert lib.yl_debug_x3_pack(w.ctypes.data_as(fp), C, M, size, raw.ctypes.data_as(Cc.c_void_p), need) == need
    units = raw.reshape(C // 16, taps, 3, 2, mpad, 8)               # [channel block][tap][piece][k-octet][filter][k]
    pieces = (units.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert not np.any(units[:, :, :, :, M:, :])                      # pad filters are zero
    # back to [m][c][tap]
    got = pieces[:, :, :, :, :M, :].transpose(4, 0, 3, 5, 1, 2).reshape(M, C, taps, 3)
    want = w.reshape(M, C, taps).astype(np.float64)
    assert np.array_equal(got.sum(axis=3), want)                     # a1 + a2 + a3 == a, exactly

    def bf16_rne(x32):
        u = x32.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

    r = w.reshape(M, C, taps).astype(np.float32)
    for pc in range(3):
        h = bf16_rne(r)
        assert np.array_equal(h.view(np.uint32), got[..., pc].astype(np.float32).view(np.uint32)), pc
        r = (r - h).astype(np.float32)
    assert np.all(np.abs(got[..., 1]) <= np.abs(got[..., 0]) * 2.0 ** -8 + 1e-45)      # each piece 8 bits below the one before


@pytest.mark.parametrize("C,M", [(16, 33), (32, 128), (48, 70), (64, 255)])
def test_row3_weight_transform_and_split_are_exact(C, M):
    """conv_f32_row3.hip's weights: the row transform of every filter row, U0 = g0, U1 = (g0 + g1 + g2) / 2,
    U2 = (g0 - g1 + g2) / 2, U3 = g2, formed in double and rounded ONCE to FP32, then as three bf16 numbers whose sum is U
    exactly; the units sit where the kernel's lanes read them ([group = (c / 16) * 3 + ky][plane][piece][k-octet][filter][k]);
    filters beyond M are zero.  And F(2,3) with these U reproduces the 3-tap correlation (float64 check of the algebra)."""
    import ctypes as Cc
    from yolo2_light_amd._lib import lib
    rng = np.random.default_rng(C * 977 + M)
    w = (rng.standard_normal((M, C, 3, 3)) * np.exp(rng.uniform(-4, 2, (M, C, 3, 3)))).astype(np.float32)
    w[rng.random(w.shape) < 0.02] = 0.0
    fp = Cc.POINTER(Cc.c_float)
    assert lib.yl_debug_row3_pack(w.ctypes.data_as(fp), C + 1, M, None, 0) < 0               # C % 16 != 0: not this kernel's layer
    need = lib.yl_debug_row3_pack(w.ctypes.data_as(fp), C, M, None, 0)
    mpad = (M + 127) // 128 * 128
    assert need == (C // 16) * 3 * 24 * mpad * 16
    raw = np.full(need // 2, 0x7fc0, dtype=np.uint16)
    assert lib.yl_debug_row3_pack(w.ctypes.data_as(fp), C, M, raw.ctypes.data_as(Cc.c_void_p), need) == need
    units = raw.reshape(C // 16, 3, 4, 3, 2, mpad, 8)                # [channel block][ky][plane][piece][k-octet][filter][k]
    assert not np.any(units[:, :, :, :, :, M:, :])                   # pad filters are zero
    pieces = (units.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    # -> [m][c][ky][plane][piece]
    got = pieces[:, :, :, :, :, :M, :].transpose(5, 0, 4, 6, 1, 2, 3).reshape(M, C, 3, 4, 3)
    g = w.astype(np.float64)                                         # [m][c][ky][kx]
    u64 = np.stack([g[..., 0], (.5 * g[..., 0] + .5 * g[..., 1]) + .5 * g[..., 2],
                    (.5 * g[..., 0] - .5 * g[..., 1]) + .5 * g[..., 2], g[..., 2]], axis=-1)
    u32 = u64.astype(np.float32)                                     # one rounding
    assert np.array_equal(got.sum(axis=4), u32.astype(np.float64))   # a1 + a2 + a3 == U, exactly

    def bf16_rne(x32):
        u = x32.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)

    r = u32.copy()
    for pc in range(3):
        h = bf16_rne(r)
        assert np.array_equal(h.view(np.uint32), got[..., pc].astype(np.float32).view(np.uint32)), pc
        r = (r - h).astype(np.float32)
    # the algebra, in float64 with the un-rounded U: Y(2t) = M0 + M1 + M2, Y(2t+1) = M1 - M2 - M3
    d = rng.standard_normal(4)
    gg = g[0, 0, 0]
    uu = u64[0, 0, 0]
    v = np.array([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
    mm = uu * v
    assert np.allclose([mm[0] + mm[1] + mm[2], mm[1] - mm[2] - mm[3]], [d[0:3] @ gg, d[1:4] @ gg], rtol=1e-12, atol=1e-12)


@pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,width,height", [("yolov3", 64, 64), ("yolov3-tiny", 96, 96), ("yolov3-spp", 64, 64),
                                               ("yolov2-voc", 96, 96), ("tiny-yolo-voc", 96, 96)])
def test_gpu_quant_rule_selects_the_reference_parsers_l_quantized(name, width, height):
    """yl_network_set_quant_rule(GPU): the INT8 layer set == `l.quantized` of the reference's own parse
    (src/additionally.c:3557-3559, 3996-4004); the default CPU rule == yolov2_forward_network_q's
    `i >= 1 && activation != LINEAR` (src/yolov2_forward_network_quantized.c:1036)."""
    cfg, wts = common.model_files(name, width, height)
    ref = refbind.RefNetwork(cfg, wts, 1, 1)
    gpu = Network.load(cfg, wts, 1, 1, quant_rule=1)
    cpu = Network.load(cfg, wts, 1, 1)
    n_gpu = n_cpu = 0
    for i in range(ref.n):
        ri = ref.layer_info(i)
        if ri["type"] != common.CONV:
            continue
        assert gpu.layer_info(i)["int8"] == (1 if ri["quantized"] else 0), "layer %d: GPU rule" % i
        want_cpu = 1 if (i >= 1 and ri["activation"] != 3) else 0
        assert cpu.layer_info(i)["int8"] == want_cpu, "layer %d: CPU rule" % i
        n_gpu += gpu.layer_info(i)["int8"]
        n_cpu += want_cpu
    assert 0 < n_gpu <= n_cpu        # the GPU rule is the stricter one (no 1x1, nothing near the [yolo] heads)
    if name.startswith("yolov3"):
        assert n_gpu < n_cpu


def test_load_weights_upto_is_tolerant_like_the_reference(tmp_path):
    """yl_network_load_weights_upto = load_weights_upto_cpu (src/additionally.c:3491): layers below the cutoff are
    read, a truncated file is not an error and leaves the remaining parameters at their initial values; the strict
    entry point refuses the same file."""
    from yolo2_light_amd._lib import lib
    cfg, wts = common.model_files("yolov3-tiny", 64, 64)
    full = Network.from_cfg(cfg, 1, 0)
    full.load_weights(wts)
    convs = [i for i, li in enumerate(full.layers()) if li["type"] == common.CONV]
    # keep the header and the first three conv layers' records only
    data = open(wts, "rb").read()
    keep = 20                                            # major, minor, revision, 64-bit seen
    for i in convs[:3]:
        li = full.layer_info(i)
        keep += 4 * (li["n"] * (4 if li["batch_normalize"] else 1) + li["n"] * li["c"] * li["size"] ** 2)
    short = str(tmp_path / "short.weights")
    open(short, "wb").write(data[:keep + 10])            # ends inside the fourth conv layer's biases
    strict = Network.from_cfg(cfg, 1, 0)
    assert lib.yl_network_load_weights(strict._h, short.encode()) == -2          # YL_ERR_IO
    part = Network.from_cfg(cfg, 1, 0)
    assert lib.yl_network_load_weights_upto(part._h, short.encode(), part.n) == 0
    for i in convs[:3]:
        assert np.array_equal(part.layer_weights(i), full.layer_weights(i)) and np.array_equal(part.layer_biases(i), full.layer_biases(i))
    assert not np.array_equal(part.layer_weights(convs[4]), full.layer_weights(convs[4]))
    # cutoff: nothing at or beyond the cutoff layer is touched even with the whole file
    cut = Network.from_cfg(cfg, 1, 0)
    assert lib.yl_network_load_weights_upto(cut._h, wts.encode(), convs[2]) == 0
    assert np.array_equal(cut.layer_weights(convs[1]), full.layer_weights(convs[1]))
    assert not np.array_equal(cut.layer_weights(convs[2]), full.layer_weights(convs[2]))
