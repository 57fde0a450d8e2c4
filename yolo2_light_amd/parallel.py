"""Image-batch sharding across the GPUs of one node + the final gather of detections.

The reference has no multi-device code.  The hot path shards
naturally: images are independent, weights are replicated, and the only
exchange is the fixed-capacity detection records produced on-device by
yl_network_detect_batch (records[B][cap][6+classes] + counts[B]).  Two forms:

  * `Group`: ONE host process, one thread + stream per device, RCCL send/recv gather on the root --
    the C-ABI's yl_group_* (csrc/group.hip), what the reference's C host binds;
  * one process per GPU under torch.distributed (bench.py's driver contract): `shard_range` for the
    split of the global batch, `gather_detections` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on
    CPU for tests) for the exchange.  The split is the library's own yl_shard_range in both forms.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import YoloHipError, check, lib


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous image range [lo, hi) owned by `rank` (GPU g gets
    images [g*B/G, (g+1)*B/G)); the first `global_batch % world` ranks get one extra."""
    first, count = C.c_int(0), C.c_int(0)
    if lib.yl_shard_range(global_batch, world, rank, C.byref(first), C.byref(count)) != 0:
        raise ValueError("bad rank/world")
    return first.value, first.value + count.value


class Group:
    """yl_group: the global batch of `model` (a prepared HOST Network, not on a device) split over `devices`."""

    def __init__(self, model, devices: Sequence[int]):
        from .network import Network
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(lib.yl_group_create(model._h, devs, len(devices), C.byref(h)), "yl_group_create")
        self._h = h
        self.n = len(devices)
        self.global_batch = model.batch
        self.classes = model.layer_info(model.n - 1)["classes"]
        self.last_outputs = model.layer_info(model.n - 1)["outputs"]
        self._dims = model.input_dims
        self._Network = Network

    def close(self) -> None:
        if self._h:
            lib.yl_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, rank: int) -> Tuple[int, int]:
        first, count = C.c_int(0), C.c_int(0)
        check(lib.yl_group_shard(self._h, rank, C.byref(first), C.byref(count)), "yl_group_shard")
        return first.value, count.value

    def member(self, rank: int):
        """borrowed replica of `rank` as a Network view (do not close it)"""
        p = lib.yl_group_member(self._h, rank)
        if not p:
            raise YoloHipError("yl_group_member failed: " + _lib.last_error())
        net = self._Network(C.c_void_p(p))
        net.close = lambda: None            # owned by the group
        net._h_keep = self
        return net

    def predict(self, images: np.ndarray) -> np.ndarray:
        w, h, c = self._dims
        x = np.ascontiguousarray(images, dtype=np.float32)
        if x.size != self.global_batch * c * h * w:
            raise ValueError("input has %d elements, group wants %d" % (x.size, self.global_batch * c * h * w))
        p = lib.yl_group_predict(self._h, x.ctypes.data_as(_lib.c_float_p))
        if not p:
            raise YoloHipError("yl_group_predict failed: " + _lib.last_error())
        return np.ctypeslib.as_array(p, shape=(self.global_batch * self.last_outputs,)).copy()

    def forward(self, inputs_dev: Optional[Sequence[int]] = None) -> None:
        arr = None
        if inputs_dev is not None:
            arr = (C.c_void_p * self.n)(*[C.c_void_p(p) for p in inputs_dev])
        check(lib.yl_group_forward(self._h, arr), "yl_group_forward")

    def synchronize(self) -> None:
        check(lib.yl_group_synchronize(self._h), "yl_group_synchronize")

    def detect_batch(self, thresh: float, nms: float, cap: int, records_dev_root: int, counts_dev_root: int) -> None:
        check(lib.yl_group_detect_batch(self._h, None, None, thresh, 1, 0, nms, cap, C.c_void_p(records_dev_root),
                                        C.c_void_p(counts_dev_root)), "yl_group_detect_batch")

    def get_boxes_batch(self, thresh: float, nms: float = 0.0, cap: int = 1024, sizes=None, relative: int = 1,
                        letter: int = 0):
        w, h = self._Network._dims(sizes, self.global_batch)
        ip = C.POINTER(C.c_int)
        rows = np.zeros((self.global_batch, cap, 6 + self.classes), dtype=np.float32)
        counts = np.zeros(self.global_batch, dtype=np.int32)
        check(lib.yl_group_get_boxes_batch(self._h, w.ctypes.data_as(ip) if w is not None else None,
                                           h.ctypes.data_as(ip) if h is not None else None, thresh, relative, letter,
                                           nms, cap, rows.ctypes.data_as(_lib.c_float_p), counts.ctypes.data_as(ip)),
              "yl_group_get_boxes_batch")
        return [rows[b, :min(int(counts[b]), cap)] for b in range(self.global_batch)], counts


def gather_detections(records, counts, group=None):
    """All-gather the per-rank record/count tensors.  records: [b, cap, row] float32,
    counts: [b] int32 (same b on every rank).  Returns ([world, b, cap, row], [world, b])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    b = records.shape[0]
    # concatenated-along-dim-0 form: accepted by both the RCCL and the gloo backends
    rec_all = torch.empty((world * b,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    cnt_all = torch.empty((world * b,), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(rec_all, records.contiguous(), group=group)
    dist.all_gather_into_tensor(cnt_all, counts.contiguous(), group=group)
    return rec_all.view((world,) + tuple(records.shape)), cnt_all.view(world, b)


def merge_detections(rec_all: np.ndarray, cnt_all: np.ndarray, cap: int) -> List[np.ndarray]:
    """Host side of the gather: per global image (rank-major order) the valid record rows.
    counts above `cap` mean the device buffer overflowed; the surplus was dropped on device."""
    world, b = cnt_all.shape
    out = []
    for r in range(world):
        for i in range(b):
            n = int(min(cnt_all[r, i], cap))
            out.append(np.array(rec_all[r, i, :n], copy=True))
    return out
