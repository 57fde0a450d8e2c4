"""Image-batch sharding across the GPUs of one node + the final gather of detections.

The reference has no multi-device code.  The hot path shards
naturally: images are independent, weights are replicated, and the only
exchange is the fixed-capacity detection records produced on-device by
yl_network_compact_detections (records[B][cap][6+classes] + counts[B]).  One
process per GPU; `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" on CPU for tests) is plumbing, the payload layout is ours.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous image range [lo, hi) owned by `rank` (GPU g gets
    images [g*B/G, (g+1)*B/G)); the first `global_batch % world` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


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
