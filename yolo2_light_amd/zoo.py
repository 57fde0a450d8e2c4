"""Model zoo: programmatic generators for the darknet ``.cfg`` files of the
configurations named in BASELINE.json, plus a tiny cfg walker used by the
synthetic-weights writer.

The reference ships these topologies as text files (``bin/yolov3.cfg``,
``bin/yolov3-tiny.cfg``, ``bin/tiny-yolo-obj_xnor.cfg``).  They are *inputs* to
the hot path, and the reference tree does not exist on the GPU box, so the
topologies are restated here as code (Darknet-53 + FPN heads is 5 residual
stages + 3 heads; tiny is a 7-conv trunk + 2 heads) and emitted as cfg text on
demand.  ``tests/test_zoo.py`` checks, whenever the reference tree is present,
that the reference's own parser builds layer-for-layer identical networks from
the generated text and from the shipped files.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

# per-layer INT8 input scales shipped in the reference cfgs ([net] input_calibration=,
# bin/yolov3.cfg:25, bin/yolov3-tiny.cfg:25): calibration data, not code.
_V3_CALIB = [15.497, 12.537] + [40] * 74
_TINY_CALIB = [15.7342, 4.41852, 9.17237, 9.70713, 13.1849, 14.9823, 15.1913,
               8.62978, 15.7353, 15.6297, 15.6939, 15.4093, 15.8055, 16]

_V2_VOC_CALIB = [15.8025, 11.6111, 10.9857, 14.9883, 11.6514, 14.9023, 15.4301, 13.8702, 15.3739, 15.584, 15.3044,
                 15.4963, 15.4139, 15.398, 15.7311, 15.2932, 15.7355, 15.2879, 5.79389, 15.6349, 15.5533, 15.453,
                 15.7935, 16]
_TINY_VOC_CALIB = [127, 3.88677, 10.5828, 10.3276, 14.3403, 15.2774, 15.2242, 8.08196, 15.7327, 16]
_V2_VOC_ANCHORS = "1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071"
_TINY_VOC_ANCHORS = "1.08,1.19,  3.42,4.41,  6.63,11.38,  9.42,5.11,  16.62,10.52"
_V3_ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"
_TINY_ANCHORS = "10,14,  23,27,  37,58,  81,82,  135,169,  344,319"
_XNOR_ANCHORS = "5.2367,6.0570, 8.2272,9.1483, 12.4093,10.7904, 9.7655,14.6023, 16.6749,16.0784"


def _fmt(v) -> str:
    return ("%g" % v) if isinstance(v, float) else str(v)


class _Cfg:
    def __init__(self) -> None:
        self.lines: List[str] = []

    def section(self, name: str, **kv) -> None:
        self.lines.append("[%s]" % name)
        for k, v in kv.items():
            self.lines.append("%s=%s" % (k, _fmt(v)))
        self.lines.append("")

    def conv(self, filters: int, size: int, stride: int = 1, bn: bool = True,
             act: str = "leaky", xnor: int = 0, bin_output: int = 0) -> None:
        kv: Dict[str, object] = {}
        if xnor:
            kv["xnor"] = 1
        if bin_output:
            kv["bin_output"] = 1
        if bn:
            kv["batch_normalize"] = 1
        kv.update(filters=filters, size=size, stride=stride, pad=1, activation=act)
        self.section("convolutional", **kv)

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"


def _net(c: _Cfg, width: int, height: int, calib=None) -> None:
    kv: Dict[str, object] = dict(batch=1, subdivisions=1, width=width, height=height, channels=3)
    if calib:
        kv["input_calibration"] = ", ".join(_fmt(float(x)) if not float(x).is_integer() else str(int(x))
                                            for x in calib)
    c.section("net", **kv)


def yolov3_cfg(width: int = 608, height: int = 608, classes: int = 80, spp: bool = False) -> str:
    """Darknet-53 backbone + 3 YOLO heads == bin/yolov3.cfg with width/height edited;
    spp=True inserts the 5/9/13 stride-1 max-pool pyramid of bin/yolov3-spp.cfg into the first head."""
    c = _Cfg()
    _net(c, width, height, _V3_CALIB)
    head_filters = 3 * (classes + 5)

    def residual(ch: int, n: int) -> None:
        for _ in range(n):
            c.conv(ch // 2, 1)
            c.conv(ch, 3)
            c.section("shortcut", **{"from": -3, "activation": "linear"})

    c.conv(32, 3)
    for ch, n in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        c.conv(ch, 3, stride=2)
        residual(ch, n)

    def head(ch: int, mask: str, with_spp: bool = False) -> None:
        for k in range(3):
            c.conv(ch, 1)
            if with_spp and k == 1:
                c.section("maxpool", stride=1, size=5)
                c.section("route", layers="-2")
                c.section("maxpool", stride=1, size=9)
                c.section("route", layers="-4")
                c.section("maxpool", stride=1, size=13)
                c.section("route", layers="-1,-3,-5,-6")
                c.conv(ch, 1)
            c.conv(ch * 2, 3)
        c.conv(head_filters, 1, bn=False, act="linear")
        c.section("yolo", mask=mask, anchors=_V3_ANCHORS, classes=classes, num=9,
                  jitter=.3, ignore_thresh=.7, truth_thresh=1, random=1)

    head(512, "6,7,8", with_spp=spp)
    c.section("route", layers="-4")
    c.conv(256, 1)
    c.section("upsample", stride=2)
    c.section("route", layers="-1, 61")
    head(256, "3,4,5")
    c.section("route", layers="-4")
    c.conv(128, 1)
    c.section("upsample", stride=2)
    c.section("route", layers="-1, 36")
    head(128, "0,1,2")
    return c.text()


def yolov3_tiny_cfg(width: int = 416, height: int = 416, classes: int = 80) -> str:
    """== bin/yolov3-tiny.cfg."""
    c = _Cfg()
    _net(c, width, height, _TINY_CALIB)
    head_filters = 3 * (classes + 5)
    for i, ch in enumerate((16, 32, 64, 128, 256, 512)):
        c.conv(ch, 3)
        c.section("maxpool", size=2, stride=2 if i < 5 else 1)
    c.conv(1024, 3)
    c.conv(256, 1)
    c.conv(512, 3)
    c.conv(head_filters, 1, bn=False, act="linear")
    c.section("yolo", mask="3,4,5", anchors=_TINY_ANCHORS, classes=classes, num=6,
              jitter=.3, ignore_thresh=.7, truth_thresh=1, random=1)
    c.section("route", layers="-4")
    c.conv(128, 1)
    c.section("upsample", stride=2)
    c.section("route", layers="-1, 8")
    c.conv(256, 3)
    c.conv(head_filters, 1, bn=False, act="linear")
    c.section("yolo", mask="1,2,3", anchors=_TINY_ANCHORS, classes=classes, num=6,
              jitter=.3, ignore_thresh=.7, truth_thresh=1, random=1)
    return c.text()


def tiny_yolo_xnor_cfg(width: int = 416, height: int = 416, classes: int = 6) -> str:
    """== bin/tiny-yolo-obj_xnor.cfg (7 XNOR convs, region head, 5 anchors)."""
    c = _Cfg()
    _net(c, width, height, None)
    c.conv(16, 3)
    c.section("maxpool", size=2, stride=2)
    for i, ch in enumerate((32, 64, 128, 256, 512)):
        c.conv(ch, 3, xnor=1, bin_output=1)
        c.section("maxpool", size=2, stride=2 if i < 4 else 1)
    c.conv(1024, 3, xnor=1, bin_output=1)
    c.conv(1024, 3, xnor=1)
    c.conv(5 * (classes + 5), 1, bn=False, act="linear")
    c.section("region", anchors=_XNOR_ANCHORS, bias_match=1, classes=classes, coords=4, num=5,
              softmax=1, jitter=.2, rescore=1, object_scale=5, noobject_scale=1, class_scale=1,
              coord_scale=1, absolute=1, thresh=.6, random=1)
    return c.text()


def yolov2_voc_cfg(width: int = 416, height: int = 416, classes: int = 20) -> str:
    """== bin/yolov2-voc.cfg (Darknet-19 trunk, passthrough reorg, region head)."""
    c = _Cfg()
    _net(c, width, height, _V2_VOC_CALIB)
    c.conv(32, 3)
    c.section("maxpool", size=2, stride=2)
    c.conv(64, 3)
    c.section("maxpool", size=2, stride=2)
    for ch in (128, 256):
        c.conv(ch, 3); c.conv(ch // 2, 1); c.conv(ch, 3)
        c.section("maxpool", size=2, stride=2)
    for ch in (512, 1024):
        c.conv(ch, 3); c.conv(ch // 2, 1); c.conv(ch, 3); c.conv(ch // 2, 1); c.conv(ch, 3)
        if ch == 512:
            c.section("maxpool", size=2, stride=2)
    c.conv(1024, 3)
    c.conv(1024, 3)
    c.section("route", layers="-9")
    c.conv(64, 1)
    c.section("reorg", stride=2)
    c.section("route", layers="-1,-4")
    c.conv(1024, 3)
    c.conv(5 * (classes + 5), 1, bn=False, act="linear")
    c.section("region", anchors=_V2_VOC_ANCHORS, bias_match=1, classes=classes, coords=4, num=5, softmax=1,
              jitter=.3, rescore=1, object_scale=5, noobject_scale=1, class_scale=1, coord_scale=1,
              absolute=1, thresh=.6, random=1)
    return c.text()


def tiny_yolo_voc_cfg(width: int = 416, height: int = 416, classes: int = 20) -> str:
    """== bin/tiny-yolo-voc.cfg."""
    c = _Cfg()
    _net(c, width, height, _TINY_VOC_CALIB)
    for i, ch in enumerate((16, 32, 64, 128, 256, 512)):
        c.conv(ch, 3)
        c.section("maxpool", size=2, stride=2 if i < 5 else 1)
    c.conv(1024, 3)
    c.conv(1024, 3)
    c.conv(5 * (classes + 5), 1, bn=False, act="linear")
    c.section("region", anchors=_TINY_VOC_ANCHORS, bias_match=1, classes=classes, coords=4, num=5, softmax=1,
              jitter=.2, rescore=1, object_scale=5, noobject_scale=1, class_scale=1, coord_scale=1,
              absolute=1, thresh=.6, random=1)
    return c.text()


MODELS = {
    "yolov3-spp": lambda w=608, h=608, **kw: yolov3_cfg(w, h, spp=True, **kw),
    "yolov3-openimages": lambda w=608, h=608, **kw: yolov3_cfg(w, h, classes=601, **kw),
    "yolov2-voc": yolov2_voc_cfg,
    "tiny-yolo-voc": tiny_yolo_voc_cfg,
    "yolov3": yolov3_cfg,
    "yolov3-tiny": yolov3_tiny_cfg,
    "tiny-yolo-xnor": tiny_yolo_xnor_cfg,
}


def write_cfg(name: str, out_dir: str, width: int, height: int, **kw) -> str:
    """Emit cfg text for `name` into out_dir and return the file path."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "%s-%dx%d.cfg" % (name, width, height))
    text = MODELS[name](width, height, **kw)
    with open(path, "w") as f:
        f.write(text)
    return path


# ---------------------------------------------------------------------------
# minimal cfg walker: conv layer shapes in file order (for the weights writer)
# ---------------------------------------------------------------------------
def parse_sections(text: str) -> List[Tuple[str, Dict[str, str]]]:
    secs: List[Tuple[str, Dict[str, str]]] = []
    for raw in text.splitlines():
        line = "".join(raw.split())
        if not line or line[0] in "#;":
            continue
        if line[0] == "[":
            secs.append((line.strip("[]"), {}))
        elif "=" in line and secs:
            k, v = line.split("=", 1)
            secs[-1][1].setdefault(k, v)
    return secs


def conv_shapes(text: str) -> List[dict]:
    """[{index, n, c, size, bn, linear, head_anchors, head_classes}] for every conv layer.

    Geometry rules follow the reference's make_*_layer (src/additionally.c:2299-2910).
    """
    secs = parse_sections(text)
    net = secs[0][1]
    h, w, c = int(net["height"]), int(net["width"]), int(net["channels"])
    outs: List[Tuple[int, int, int]] = []
    convs: List[dict] = []
    for idx, (typ, o) in enumerate(secs[1:]):
        if typ == "convolutional":
            n, size, stride = int(o.get("filters", 1)), int(o.get("size", 1)), int(o.get("stride", 1))
            pad = size // 2 if int(o.get("pad", 0)) else int(o.get("padding", 0))
            convs.append(dict(index=idx, n=n, c=c, size=size, bn=int(o.get("batch_normalize", 0)),
                              linear=o.get("activation", "logistic") == "linear",
                              head_anchors=0, head_classes=0, before_shortcut=False))
            h, w, c = (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1, n
        elif typ == "maxpool":
            stride = int(o.get("stride", 1))
            size = int(o.get("size", stride))
            pad = int(o.get("padding", size - 1))
            h, w = (h + pad - size) // stride + 1, (w + pad - size) // stride + 1
        elif typ == "upsample":
            s = int(o.get("stride", 2))
            h, w = h * s, w * s
        elif typ == "route":
            ids = [int(x) for x in o["layers"].split(",")]
            ids = [i + idx if i < 0 else i for i in ids]
            h, w = outs[ids[0]][0], outs[ids[0]][1]
            c = sum(outs[i][2] for i in ids)
        elif typ == "reorg":
            s = int(o.get("stride", 1))
            h, w, c = h // s, w // s, c * s * s
        elif typ in ("yolo", "region"):
            # tag the linear conv feeding this head so the weights writer can bias objectness
            classes = int(o.get("classes", 20))
            if typ == "yolo":
                anchors = len(o["mask"].split(",")) if "mask" in o else int(o.get("num", 1))
            else:
                anchors = int(o.get("num", 1))
            if convs and convs[-1]["index"] == idx - 1:
                convs[-1]["head_anchors"] = anchors
                convs[-1]["head_classes"] = classes
        elif typ == "shortcut":
            # shortcut keeps h, w, c; tag the conv producing the residual branch
            if convs and convs[-1]["index"] == idx - 1:
                convs[-1]["before_shortcut"] = True
        outs.append((h, w, c))
    return convs
