"""The N>1 path (image-batch sharding + gather of fixed-capacity detection
records) on CPU with world_size 2 over gloo; the group API's argument checks without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common  # noqa: F401
from yolo2_light_amd import parallel


def test_shard_range_partitions_batch():
    for gb in (1, 7, 64, 65, 128):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, cap, row = 3, 8, 6 + 4
    rng = np.random.default_rng(100 + rank)
    counts = torch.tensor(rng.integers(0, cap + 3, size=b), dtype=torch.int32)     # may exceed cap (overflow)
    rec = torch.zeros((b, cap, row), dtype=torch.float32)
    for i in range(b):
        n = min(int(counts[i]), cap)
        rec[i, :n] = torch.tensor(rng.random((n, row)), dtype=torch.float32) + rank * 10
    rec_all, cnt_all = parallel.gather_detections(rec, counts)
    merged = parallel.merge_detections(rec_all.numpy(), cnt_all.numpy(), cap)
    q.put((rank, rec.numpy(), counts.numpy(), [m.copy() for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_detections_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b, cap = 3, 8
    # every rank sees the same merged list: rank-major, image-minor, rows == the owner's valid rows
    for rank, _, _, merged in results:
        assert len(merged) == world * b
        for owner, rec, counts, _ in results:
            for i in range(b):
                n = min(int(counts[i]), cap)
                got = merged[owner * b + i]
                assert got.shape[0] == n
                assert np.array_equal(got, rec[i, :n])


def test_group_without_gpu_fails_loudly():
    """yl_group_create has no CPU fallback either; argument errors are reported before any device work"""
    import ctypes as C
    from yolo2_light_amd import Network, YoloHipError
    from yolo2_light_amd._lib import lib
    cfg, wts = common.model_files("yolov3-tiny", 96, 96)
    model = Network.load(cfg, wts, 4, 0)
    if lib.yl_device_count() == 0:
        with pytest.raises(YoloHipError):
            parallel.Group(model, [0])
    with pytest.raises(YoloHipError):
        parallel.Group(model, [])                       # no devices
    f, c = C.c_int(), C.c_int()
    assert lib.yl_shard_range(64, 8, 8, C.byref(f), C.byref(c)) != 0
    assert lib.yl_shard_range(10, 4, 1, C.byref(f), C.byref(c)) == 0 and (f.value, c.value) == (3, 3)
