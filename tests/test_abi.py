"""The C-ABI library loads and exports every symbol include/*.h declares -- the drop-in boundary (yolo2_hip.h) and the
test / lab instrumentation (yolo2_hip_lab.h) -- (no compute calls: this runs without a GPU)."""
import os
import re

import common  # noqa: F401
from yolo2_light_amd import _lib

HEADER = os.path.join(common.ROOT, "include", "yolo2_hip.h")
LAB_HEADER = os.path.join(common.ROOT, "include", "yolo2_hip_lab.h")


def header_functions(paths=(HEADER, LAB_HEADER)):
    names = []
    for path in paths:
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(yl_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_the_drop_in_header_declares_no_test_hooks():
    """the yl_debug_* hooks are lab instrumentation, not part of the boundary a caller sees"""
    assert not [n for n in header_functions((HEADER,)) if n.startswith("yl_debug")]
    assert [n for n in header_functions((LAB_HEADER,)) if n.startswith("yl_debug")]


def test_every_declared_symbol_is_exported():
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), "libyolo2hip.so does not export %s" % n


def test_binding_table_matches_header():
    assert sorted(_lib.EXPORTED) == header_functions()


def test_abi_version_marker():
    """header, library and binding agree on YL_ABI_VERSION (a changed signature behind an unchanged symbol name is
    caught at load time)"""
    m = re.search(r"#define\s+YL_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m and int(m.group(1)) == _lib.ABI_VERSION == _lib.lib.yl_abi_version()


def test_no_gpu_means_loud_failure_not_fallback():
    """Device entry points must fail with an error (never compute on the CPU)."""
    import numpy as np
    import descs as D
    if _lib.lib.yl_device_count() > 0:
        return
    w = np.ones(3 * 4, np.float32)
    d = D.conv(1, 4, 4, 3, 4, 1, 1, 0, D.LINEAR, w, np.zeros(4, np.float32))
    from yolo2_light_amd import Network, YoloHipError
    net = Network.from_desc([d], 1, 4, 4, 3)
    try:
        net.to_device(0)
    except YoloHipError as e:
        assert "no HIP device" in str(e) or "hip" in str(e).lower()
    else:
        raise AssertionError("to_device succeeded without a GPU")
    try:
        net.predict(np.zeros((1, 3, 4, 4), np.float32))
    except YoloHipError:
        pass
    else:
        raise AssertionError("predict succeeded without a GPU")
    # the on-device prep passes have no host fallback either (the host passes are separate entry points)
    cfg, wts = __import__("common").model_files("yolov3-tiny", 64, 64)
    net2 = Network.from_cfg(cfg, 1, 0)
    net2.load_weights(wts)
    assert _lib.lib.yl_network_prepare_on_device(net2._h, 0) == -4        # YL_ERR_DEVICE
    assert net2.layer_info(0)["batch_normalize"] == 1                     # nothing was folded on the host instead


def test_layer_desc_struct_matches_header_field_order():
    text = open(HEADER).read()
    body = text[text.index("typedef struct yl_layer_desc {"):text.index("} yl_layer_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int8_t|int|float|unsigned char)\s*", "", decl)
        for part in decl.split(","):
            fields.append(part.replace("*", "").replace("const", "").strip())
    assert fields == [f[0] for f in _lib.LayerDesc._fields_]
