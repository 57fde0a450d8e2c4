"""Builders for yl_layer_desc arrays (single-op and small networks) used by the
op-level parity tests through yl_network_create_from_desc."""
from __future__ import annotations

import ctypes as C

import numpy as np

from yolo2_light_amd import _lib
from yolo2_light_amd._lib import LayerDesc

CONV, MAXPOOL, ROUTE, SHORTCUT, REGION, YOLO, UPSAMPLE, REORG = 0, 3, 8, 13, 21, 22, 23, 24
LINEAR, LEAKY = 3, 7

_keep = []   # keep numpy buffers alive while descs reference them


def _fp(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    _keep.append(a)
    return a.ctypes.data_as(_lib.c_float_p)


def _ip(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    _keep.append(a)
    return a.ctypes.data_as(_lib.c_int_p)


def conv(batch, w, h, c, n, size, stride, pad, act, weights, biases, xnor=0, weights_int8=None, in_mult=0.0,
         w_mult=0.0, mean_arr=None):
    d = LayerDesc()
    d.type = CONV; d.activation = act
    d.batch = batch; d.w = w; d.h = h; d.c = c; d.n = n
    d.size = size; d.stride = stride; d.pad = pad
    d.out_h = (h + 2 * pad - size) // stride + 1
    d.out_w = (w + 2 * pad - size) // stride + 1
    d.out_c = n
    d.outputs = d.out_h * d.out_w * n
    d.inputs = w * h * c
    d.xnor = xnor
    d.weights = _fp(weights)
    d.biases = _fp(biases)
    if weights_int8 is not None:
        a = np.ascontiguousarray(weights_int8, dtype=np.int8)
        _keep.append(a)
        d.weights_int8 = a.ctypes.data_as(_lib.c_int8_p)
        d.input_quant_multipler = in_mult
        d.weights_quant_multipler = w_mult
    if mean_arr is not None:
        d.mean_arr = _fp(mean_arr)
    return d


def maxpool(batch, w, h, c, size, stride, pad=None):
    if pad is None:
        pad = size - 1
    d = LayerDesc()
    d.type = MAXPOOL; d.activation = LINEAR
    d.batch = batch; d.w = w; d.h = h; d.c = c
    d.size = size; d.stride = stride; d.pad = pad
    d.out_w = (w + pad - size) // stride + 1
    d.out_h = (h + pad - size) // stride + 1
    d.out_c = c
    d.outputs = d.out_h * d.out_w * c
    d.inputs = w * h * c
    return d


def upsample(batch, w, h, c, stride, scale=1.0):
    d = LayerDesc()
    d.type = UPSAMPLE; d.activation = LINEAR
    d.batch = batch; d.w = w; d.h = h; d.c = c; d.stride = stride; d.scale = scale
    d.out_w = w * stride; d.out_h = h * stride; d.out_c = c
    d.outputs = d.out_w * d.out_h * c
    d.inputs = w * h * c
    return d


def shortcut(batch, index, from_dims, cur_dims, act=LINEAR):
    """from_dims = (w,h,c) of layer `index`; cur_dims = (w,h,c) of the running input."""
    d = LayerDesc()
    d.type = SHORTCUT; d.activation = act
    d.batch = batch; d.index = index
    d.w, d.h, d.c = from_dims
    d.out_w, d.out_h, d.out_c = cur_dims
    d.outputs = cur_dims[0] * cur_dims[1] * cur_dims[2]
    d.inputs = d.outputs
    return d


def route(batch, input_layers, input_sizes, out_dims):
    d = LayerDesc()
    d.type = ROUTE; d.activation = LINEAR
    d.batch = batch; d.n = len(input_layers)
    d.input_layers = _ip(input_layers)
    d.input_sizes = _ip(input_sizes)
    d.outputs = int(sum(input_sizes)); d.inputs = d.outputs
    d.out_w, d.out_h, d.out_c = out_dims
    d.w, d.h, d.c = out_dims
    return d


def yolo(batch, w, h, n, classes, total, mask, anchors):
    d = LayerDesc()
    d.type = YOLO; d.activation = LINEAR
    d.batch = batch; d.w = w; d.h = h; d.n = n; d.classes = classes; d.total = total; d.coords = 4
    d.c = n * (classes + 5)
    d.out_w = w; d.out_h = h; d.out_c = d.c
    d.outputs = w * h * d.c; d.inputs = d.outputs
    d.mask = _ip(mask)
    d.anchors = _fp(anchors)
    return d


def region(batch, w, h, n, classes, anchors, softmax=1):
    d = LayerDesc()
    d.type = REGION; d.activation = LINEAR
    d.batch = batch; d.w = w; d.h = h; d.n = n; d.classes = classes; d.coords = 4; d.total = n
    d.c = n * (classes + 5); d.softmax = softmax
    d.outputs = w * h * d.c; d.inputs = d.outputs
    d.anchors = _fp(anchors)
    return d


def reorg(batch, w, h, c, stride):
    d = LayerDesc()
    d.type = REORG; d.activation = LINEAR
    d.batch = batch; d.w = w; d.h = h; d.c = c; d.stride = stride
    d.out_w = w // stride; d.out_h = h // stride; d.out_c = c * stride * stride
    d.outputs = d.out_w * d.out_h * d.out_c; d.inputs = w * h * c
    return d
