"""Synthetic darknet ``.weights`` writer.

No pretrained weights exist offline, and without a weights file the
reference leaves BN variance at 0 so fusing multiplies weights by 1e6
(src/additionally.c:87,2809).  Every parity and bench run therefore uses a
deterministic synthetic file in exactly the format
``load_weights_upto_cpu`` / ``load_convolutional_weights_cpu`` read
(src/additionally.c:3491-3529, 3459-3468):

    int32 major, minor, revision ; uint64 seen (major*10+minor >= 2)
    per CONVOLUTIONAL layer, in order:
        biases[n] ; if batch_normalize: scales[n], rolling_mean[n], rolling_variance[n]
        weights[n*c*size*size]                     all float32 little-endian

The statistics are chosen so activations stay O(1) through 75 conv layers and 23
residual adds (a net whose activations overflow exercises neither the kernels'
numerics nor -- through DVFS -- their real power/clock behaviour):
  w ~ N(0, sqrt(2/K)) (He), bias ~ N(0, .1), mean ~ N(0, .1), var ~ U(.5, 1.5),
  scales ~ g * U(.5, 1.5) with g = 0.917 (makes E[scale^2/var] = 1, variance
  preserving through conv+BN+leaky) and g = 0.25 for the conv that feeds a
  [shortcut] (the residual branch adds ~6 % variance per block instead of 100 %).
"""
from __future__ import annotations

import struct

import numpy as np

from .zoo import conv_shapes

_G_PLAIN = 0.917
_G_RESIDUAL = 0.25


def write_synthetic_weights(cfg_text: str, path: str, seed: int = 1, obj_bias: float = -3.0,
                            cls_bias: float = 0.0, head_bias_delta=None) -> int:
    """Write a synthetic weights file for the network described by cfg_text.

    The objectness channel of every detection-head conv gets `obj_bias` added so
    that only some boxes pass the 0.24 threshold; `cls_bias` is added to the class channels
    (a trained detector is confident about one or two classes per box, an untrained head
    about half of them); `head_bias_delta` = one float array per detection-head conv (in cfg
    order) added to its biases after they were drawn -- bench.py uses it to give the random
    head a detector-like output density.  Returns the number of float32 values written.
    """
    rng = np.random.default_rng(seed)
    total = 0
    head_i = 0
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", 0, 2, 0))
        f.write(struct.pack("<Q", 0))
        for cv in conv_shapes(cfg_text):
            n, c, size = cv["n"], cv["c"], cv["size"]
            k = c * size * size
            bias = rng.normal(0.0, 0.1, n).astype(np.float32)
            if cv["head_anchors"]:
                per = 5 + cv["head_classes"]
                for a in range(cv["head_anchors"]):
                    if a * per + 4 < n:
                        bias[a * per + 4] += np.float32(obj_bias)
                    if cls_bias:
                        bias[a * per + 5:min((a + 1) * per, n)] += np.float32(cls_bias)
                if head_bias_delta is not None and head_i < len(head_bias_delta):
                    bias += np.asarray(head_bias_delta[head_i], dtype=np.float32).reshape(n)
                head_i += 1
            bias.astype("<f4").tofile(f)
            total += n
            if cv["bn"]:
                g = _G_RESIDUAL if cv["before_shortcut"] else _G_PLAIN
                (g * rng.uniform(0.5, 1.5, n)).astype("<f4").tofile(f)
                rng.normal(0.0, 0.1, n).astype("<f4").tofile(f)
                rng.uniform(0.5, 1.5, n).astype("<f4").tofile(f)
                total += 3 * n
            w = rng.normal(0.0, np.sqrt(2.0 / k), n * k).astype("<f4")
            w.tofile(f)
            total += n * k
    return total
