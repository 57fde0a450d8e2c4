#!/usr/bin/env python
"""bench.py -- images/sec of the YOLO inference hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched with torch.distributed.run, one rank per GPU (RCCL).
A "step" = one pass of the hot path over one batch of synthetic images that
are already resident in HBM: forward of every layer (FP32 MFMA conv + small
layers) -> on-device detection decode + compaction + per-class NMS for every image
(yl_network_detect_batch) -> (N>1) RCCL all-gather of the fixed-capacity
detection records to every rank.
Default workload = BASELINE.json's metric config: yolov3.cfg 608x608,
batch 64 per GPU, FP32, synthetic weights/images (no datasets or checkpoints
exist offline).  Scaling is weak: the path shards by independent images, each
rank owns `--batch` images and the replicated weights.

Rank 0 prints ONE JSON line with the driver's fields plus
  "roofline"     -- dominant kernel (the FP32 MFMA implicit-GEMM conv instance that
                    carries most FLOPs): algorithmic FLOPs of its launches / their
                    HIP-event-measured duration inside the timed region, vs the
                    157.3 TFLOP/s FP32-matrix peak (MI355X_MICROARCH.md)
  "cpu_baseline" -- the reference's own CPU path (oracle/_ref, its fastest documented
                    build AVX+OpenMP) timed on this host, N=1 / rank 0 only.
  "pcie_inclusive" -- the same workload host-to-host (u8 frames in, detection rows out; and
                    yl_network_predict on float host images), N=1 only; reported beside
                    `value`, never as `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="yolov3", choices=["yolov3", "yolov3-tiny", "tiny-yolo-xnor"])
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--mode", default="fp32", choices=["fp32", "int8"])
    ap.add_argument("--thresh", type=float, default=0.24)
    ap.add_argument("--cap", type=int, default=1024, help="detection records per image")
    ap.add_argument("--nms", type=float, default=0.4, help="do_nms_sort threshold (src/main.c:173); 0 = compaction only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive leg after the timed region")
    ap.add_argument("--raw-head", action="store_true",
                    help="keep the uncalibrated random detection head (thousands of boxes per image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--layers", action="store_true", help="also print a per-layer table to stderr")
    ap.add_argument("--tile", type=int, default=0, help="force K1 tile config (tuning)")
    ap.add_argument("--i8-tile", type=int, default=0, help="force K2 (INT8) tile config (tuning)")
    ap.add_argument("--no-fuse", action="store_true", help="keep [shortcut] layers as separate kernels")
    return ap.parse_args()


def cpu_baseline(cfg: str, wts: str, width: int, height: int, quantized: int, budget_s: float):
    """Reference CPU path (oracle/_ref/libyolo2ref_fast.so: `make AVX=1 OPENMP=1` flags) on B=1."""
    from oracle import refbind
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    fast = refbind.available(fast=True) and (" avx2 " in flags + " ") and (" fma " in flags + " ")
    if not fast and not refbind.available():
        return None
    ref = refbind.RefNetwork(cfg, wts, 1, quantized, fast=fast)
    x = np.random.default_rng(2222222).random((1, 3, height, width), dtype=np.float32)
    t_warm = ref.time_predict(x, 1)
    iters = max(1, min(20, int(budget_s / max(t_warm, 1e-3))))
    t = ref.time_predict(x, iters)
    cores = os.cpu_count() if fast else 1
    return {
        "value": iters / t, "unit": "images/sec", "cores": cores, "kind": "reference",
        "sample": "%d image(s) of the same workload at batch 1 through network_predict_%s of the reference "
                  "(%s build), wall clock %.2f s" % (iters, "quantized" if quantized else "cpu",
                                                     "AVX=1 OPENMP=1" if fast else "scalar -O2", t),
    }


def calibrate_head(Network, cfg: str, wts: str, size: int, device: int, thresh: float, quantized: int):
    """Bias shifts that give the random-weight YOLO heads a detector-like output density.
    With i.i.d. weights whole anchor channels saturate (every cell of an anchor passes the
    objectness threshold with ~40 of 80 classes each: ~3800 boxes/image at 608), which no trained
    detector produces and which would turn the post-processing stage into the benchmark.  A
    2-image probe measures each head channel's logit distribution; the objectness channels are
    shifted so that 0.3 % of the cells pass `thresh` (~70 boxes per 608x608 image, COCO-like) and
    the class channels so that ~1.5 % of (box, class) pairs exceed 0.5.  Convolution work is
    unchanged (same shapes, same FLOPs); only head biases move."""
    probe = Network.load(cfg, wts, 2, quantized, device=device)
    x = np.random.default_rng(11).random((2, 3, size, size), dtype=np.float32)
    probe.predict(x)
    deltas = []
    for i in range(probe.n):
        li = probe.layer_info(i)
        if li["type"] != 22:        # YL_YOLO
            continue
        per = 5 + li["classes"]
        o = probe.layer_output(i - 1).reshape(2, li["n"], per, -1)
        d = np.zeros((li["n"], per), dtype=np.float32)
        target_obj = float(np.log(thresh / (1.0 - thresh)))
        for a in range(li["n"]):
            d[a, 4] = target_obj - np.quantile(o[:, a, 4, :], 0.997)
            for c in range(5, per):
                d[a, c] = 0.0 - np.quantile(o[:, a, c, :], 0.985)
        deltas.append(d.reshape(-1))
    probe.close()
    return deltas


def pcie_inclusive(net, torch, stream, args, rec, cnt, steps: int = 3):
    """Host-to-host rate of the same workload (reported next to `value`, never as `value`):
    (a) decoder-style input: B u8 768x576x3 frames in pageable host memory per step ->
        yl_network_set_input_u8 (pinned staging, H2D, GPU /255 + resize_image) -> forward ->
        batched detections + NMS on the GPU -> rows and counts to pinned host memory; steps are
        pipelined on one stream, one sync at the end;
    (b) the reference's own boundary: yl_network_predict(float CHW host images), which also
        brings the head tensors back (what network_predict_cpu leaves in l.output)."""
    B = args.batch
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 256, size=(576, 768, 3), dtype=np.uint8) for _ in range(8)]
    rec_h = torch.empty(rec.shape, dtype=rec.dtype).pin_memory()
    cnt_h = torch.empty(cnt.shape, dtype=cnt.dtype).pin_memory()

    def one():
        for b in range(B):
            net.set_input_u8(b, frames[b % len(frames)])
        net.forward_staged()
        with torch.cuda.stream(stream):
            net.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr(),
                             sizes=(768, 576), relative=0)
            rec_h.copy_(rec, non_blocking=True)
            cnt_h.copy_(cnt, non_blocking=True)

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    t_u8 = (time.perf_counter() - t0) / steps
    x_host = np.random.default_rng(8).random((B, 3, args.size, args.size), dtype=np.float32)
    net.predict(x_host)
    t0 = time.perf_counter()
    for _ in range(2):
        net.predict(x_host)
    t_f32 = (time.perf_counter() - t0) / 2
    return {
        "value": B / t_u8, "unit": "images/sec", "ms_per_step": t_u8 * 1e3,
        "what": "u8 768x576x3 host frames -> yl_network_set_input_u8 -> forward -> yl_network_detect_batch -> "
                "rows+counts on the host; %d pipelined steps" % steps,
        "predict_float_host": {"value": B / t_f32, "unit": "images/sec", "ms_per_step": t_f32 * 1e3,
                               "what": "yl_network_predict(float CHW host batch): pinned staging + H2D of "
                                       "%.0f MB + forward + D2H of the heads" % (x_host.nbytes / 1e6)},
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set) -> always use the RCCL path, even at world 1,
    # so the single-GPU box exercises exactly the code the multi-GPU runs execute
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from yolo2_light_amd import Network, weights, zoo
    from yolo2_light_amd._lib import lib

    quantized = 1 if args.mode == "int8" else 0
    work = tempfile.mkdtemp(prefix="yl_bench_r%d_" % rank)
    cfg = zoo.write_cfg(args.model, work, args.size, args.size)
    wts = os.path.join(work, "synthetic.weights")
    with open(cfg) as f:
        weights.write_synthetic_weights(f.read(), wts, seed=1)
    if not args.raw_head:
        with open(cfg) as f:
            cfg_text = f.read()
        deltas = calibrate_head(Network, cfg, wts, args.size, local_rank, args.thresh, quantized)
        if deltas:
            weights.write_synthetic_weights(cfg_text, wts, seed=1, head_bias_delta=deltas)
    net = Network.load(cfg, wts, args.batch, quantized, device=local_rank, fuse=not args.no_fuse)
    # one explicit (non-default) HIP stream shared by our kernels and torch/RCCL so the
    # compaction -> all-gather dependency is ordinary stream order
    stream = torch.cuda.Stream(device=dev)
    net.set_stream(stream.cuda_stream)
    if args.tile:
        net.set_conv_tile(args.tile)
    if args.i8_tile:
        net.set_int8_tile(args.i8_tile)

    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(2222222 + rank)
    x = torch.rand((B, 3, args.size, args.size), generator=gen, device=dev, dtype=torch.float32)
    last = net.layer_info(net.n - 1)
    classes = last["classes"]
    rec = torch.zeros((B, args.cap, 6 + classes), device=dev, dtype=torch.float32)
    cnt = torch.zeros((B,), device=dev, dtype=torch.int32)
    if use_dist:
        rec_all = torch.zeros((world * B, args.cap, 6 + classes), device=dev, dtype=torch.float32)
        cnt_all = torch.zeros((world * B,), device=dev, dtype=torch.int32)

    n_layers = net.n
    layer_ms = np.zeros(n_layers, dtype=np.float64)

    MAX_SLOTS = 64      # HIP-event timing slots: one per timed step (wraps beyond 64 steps)

    def step(slot: int):
        with torch.cuda.stream(stream):
            net.forward_timed(x.data_ptr(), slot)   # HIP events around every layer, no host sync
            if args.nms > 0:    # get_network_boxes + do_nms_sort for every image, on the GPU
                net.detect_batch(args.thresh, args.nms, args.cap, rec.data_ptr(), cnt.data_ptr())
            else:
                net.compact_detections(args.thresh, args.cap, rec.data_ptr(), cnt.data_ptr())
            if use_dist:
                dist.all_gather_into_tensor(rec_all, rec)
                dist.all_gather_into_tensor(cnt_all, cnt)

    for _ in range(args.warmup):
        step(0)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k % MAX_SLOTS)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # per-kernel durations measured inside the timed region (read after it ended)
    n_slots = min(args.steps, MAX_SLOTS)
    for sl in range(n_slots):
        ms, _ = net.layer_times(sl)
        layer_ms[:] += ms
    layer_ms *= args.steps / n_slots        # layer_ms holds the sum over all timed steps
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- roofline of the dominant kernel (rank-local measurement) -------------
    infos = net.layers()
    kern = {}
    for i, li in enumerate(infos):
        if li["type"] != 0:
            continue
        name = net.layer_kernel(i)
        flops = 2.0 * li["n"] * li["size"] ** 2 * li["c"] * li["out_h"] * li["out_w"] * B
        k = kern.setdefault(name, {"flops": 0.0, "ms": 0.0, "launches": 0})
        k["flops"] += flops
        k["ms"] += layer_ms[i] / args.steps
        k["launches"] += 1
    dom_name = max(kern, key=lambda n: kern[n]["flops"])
    dom = kern[dom_name]
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    conv_ms = sum(k["ms"] for k in kern.values())
    conv_flops = sum(k["flops"] for k in kern.values())
    other_ms = float(layer_ms.sum() / args.steps - conv_ms)
    # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes (they
    # serialise kernels), so the per-launch figure comes from the committed summary of those
    # passes for this exact workload (profiles/pmc_traffic.json), null for any other workload
    traffic = None
    algo_bytes = None
    try:
        if args.model == "yolov3" and args.size == 608 and B == 64 and args.mode == "fp32":
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f).get(dom_name)
            if pt:
                traffic = (2.0 * pt["fetch_kib"] + pt["write_kib"]) * 1024.0
    except (OSError, ValueError):
        traffic = None
    # algorithmic bytes per launch of the dominant kernel: input + weights read once, output written once
    ab = 0.0
    for i, li in enumerate(infos):
        if li["type"] == 0 and net.layer_kernel(i) == dom_name:
            ab += 4.0 * (B * li["c"] * li["h"] * li["w"] + li["n"] * li["c"] * li["size"] ** 2
                         + B * li["n"] * li["out_h"] * li["out_w"])
    algo_bytes = ab / max(dom["launches"], 1)
    # post-processing stage on its own (after the timed region): HIP events on the same stream
    det_counts = cnt.cpu().numpy()
    with torch.cuda.stream(stream):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            if args.nms > 0:
                net.detect_batch(args.thresh, args.nms, args.cap, rec.data_ptr(), cnt.data_ptr())
            else:
                net.compact_detections(args.thresh, args.cap, rec.data_ptr(), cnt.data_ptr())
        e1.record(stream)
    torch.cuda.synchronize()
    detect_ms = e0.elapsed_time(e1) / 5
    roofline = {
        "bound": "mfma", "kernel": dom_name,
        "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / FP32_MATRIX_PEAK_TFLOPS,
        "traffic": traffic, "traffic_unit": "bytes/launch (PMC, separate passes)",
        "algorithmic_bytes_per_launch": algo_bytes,
        "algorithmic_flops_per_launch": dom["flops"] / max(dom["launches"], 1),
        "launches_per_step": dom["launches"],
        "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
        "all_conv_tflops": conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
        "conv_ms_per_step": conv_ms, "other_layers_ms_per_step": other_ms,
        # Winograd F(2x2,3x3) issues 16 multiplies per 2x2 output tile instead of 36: `achieved`
        # stays the ALGORITHMIC rate (2*M*K*N per launch / duration) and can exceed the
        # matrix peak; `issued_mfma_tflops` is what the MFMA pipe actually executed
        "issued_mfma_tflops": (achieved / 2.25 if "wino" in dom_name else achieved),
        "detect_ms_per_step": detect_ms,
        "detections_per_image": {"mean": float(det_counts.mean()), "max": int(det_counts.max())},
        "by_kernel": {n: {"tflops": (k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0),
                          "ms_per_step": k["ms"], "launches": k["launches"]} for n, k in kern.items()},
    }

    if rank == 0:
        if args.layers:
            for i, li in enumerate(infos):
                print("%3d type=%2d %-28s %8.3f ms" % (i, li["type"], net.layer_kernel(i), layer_ms[i] / args.steps),
                      file=sys.stderr)
        cpu = None
        e2e = None
        if world == 1 and not args.no_e2e:
            try:
                e2e = pcie_inclusive(net, torch, stream, args, rec, cnt)
            except Exception as ex:      # the headline number must not depend on this leg
                e2e = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(cfg, wts, args.size, args.size, quantized, args.cpu_seconds)
            except Exception as e:      # the baseline is reported, never required
                cpu = {"error": repr(e)}
        total_images = world * B * args.steps
        out = {
            "metric": "images/sec (whole node) %s %dx%d batch %d %s" % (args.model, args.size, args.size, B,
                                                                       args.mode.upper()),
            "value": total_images / elapsed,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.mode == "fp32" else "i8", "data": "synthetic",
            "config": {"workload": "%s.cfg %dx%d batch=%d/GPU %s, synthetic weights+images resident in HBM, "
                                   "forward + on-device detection decode/compaction + NMS%s" % (
                                       args.model, args.size, args.size, B, args.mode.upper(),
                                       " + RCCL all-gather of detections" if use_dist else ""),
                       "global_batch": world * B, "parallelism": "image-batch sharding x%d" % world,
                       "gflop_per_image": net.flops_per_image / 1e9},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pcie_inclusive": e2e,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
