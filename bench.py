#!/usr/bin/env python
"""bench.py -- images/sec of the YOLO inference hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 runs one rank per GPU (RCCL).  Under torch.distributed.run (RANK set: the driver's form) this process IS
  a rank and WORLD_SIZE must equal N; started plainly with N > 1 the script launches the N ranks itself
  (relaunch_one_rank_per_gpu) and fails loudly when the node has fewer than N devices -- `--gpus N` never times
  fewer than N GPUs.
A "step" = one pass of the hot path over one batch of synthetic images that are already resident in
HBM: forward of every layer -> on-device detection decode + compaction + per-class NMS for every image
(yl_network_detect_batch) -> (N>1) RCCL all-gather of the fixed-capacity detection records.

Default workload = BASELINE.json's metric config: yolov3.cfg 608x608, GLOBAL batch 64, synthetic
weights/images (no datasets or checkpoints exist offline).  The metric is "FP32 & INT8": `value` is the FP32
leg, the `-quantized` INT8 leg of the same workload is timed in the same run and reported under "int8".
Scaling is STRONG by default (config 3: the batch of 64 independent images is sharded over the N GPUs with
yl_shard_range, 64/N images per GPU); `--scaling weak` keeps `--batch` images per GPU.

Rank 0 prints ONE compact JSON line (< 4 KB: compact_line) as the LAST stdout line -- the driver's fields, `roofline`,
`cpu_baseline` and one number per side leg -- and writes everything else to bench_detail.json + stderr:
  "roofline"     -- dominant kernel of the FP32 leg.  K1r / K1x run on the BF16 matrix pipe with every FP32 operand the
                    exact sum of three bf16 pieces: `achieved` = ISSUED BF16-MFMA FLOPs of its launches / their
                    HIP-event-measured duration inside the timed region, `peak` = the 2 500 TFLOP/s dense BF16 peak,
                    `frac` <= 1.  `algorithmic_tflops` (2*M*K*N) and its ratio to the 157.3 TFLOP/s
                    FP32-matrix peak are separate fields.  `traffic` is MEASURED in the run (N=1, default
                    line): two child runs of one step under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE
                    (separate passes); the committed passes (profiles/pmc_traffic.json) are the fallback and the
                    source for the other legs.
  "torchrun_world1" -- the driver's multi-GPU launch form (torch.distributed.run, RCCL all-gather in the step) at
                    world size 1 on this box, so that path executes on every default run (N=1 only).
  "int8"         -- value / ms_per_step of the INT8 leg, its roofline against HBM (algorithmic bytes of the
                    dominant INT8 kernel's launches / their measured duration vs 8 TB/s) with the INT8-MFMA
                    rate beside it, and the INT8-vs-FP32 detection agreement on the same images.
  "cpu_baseline" -- the reference's own CPU path (oracle/_ref, its fastest documented build AVX+OpenMP)
                    timed on this host, N=1 / rank 0 only.
  "batch_sweep"  -- FP32 images/sec of ONE GPU at 8/16/32 images per step: what a rank of a strong-scaled
                    8/4/2-GPU run has to do (N=1 only).
  "group_n1"     -- the same step through the single-process multi-GPU C-ABI (yl_group_*, RCCL send/recv
                    gather) on the GPU of this process (N=1 only).
  "pcie_inclusive" -- the same workload host-to-host; reported beside `value`, never as `value`.
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

# /opt/skills/guides/MI355X_MICROARCH.md
FP32_MATRIX_PEAK_TFLOPS = 157.3     # "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
FP32_VECTOR_PEAK_TFLOPS = 157.3     # "Peak FP32 (vector)": v_fma_f32 (v_pk_fma_f32), the same 64 FLOP/clk/SIMD -- its own constant
NOMINAL_SCLK_MHZ = 2400.0           # the clock that peak is quoted at
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA
INT8_MFMA_PEAK_TOPS = 5000.0        # i8 MFMA = 2x the bf16 rate (~2.5 PF dense): 2048 op/clk/SIMD; ubench 4404
HBM_PEAK_GBS = 8000.0               # HBM3E spec; 6.29 TB/s measured for a float4 copy
# XNOR roof, from measured VALU issue rates (tools/valu_issue_bench.hip -> profiles/r4_valu_issue_bench.txt; clk at 2.4 GHz
# per wave64 instruction and SIMD, 8 waves per SIMD, independent accumulators, SGPR weights): v_fma_f32 / v_add_u32 /
# v_xor_b32 / v_and_b32 2.3-2.6 (the guide's 2-cycle SIMD-32 rate), v_xnor_b32 4.3 and v_bcnt_u32_b32 4.2 -- HALF-rate
# instructions.  Round 3's kernel (v_xnor + accumulating v_bcnt per 32 bit-MACs and lane) had a nominal roof of
# 1024 SIMDs x 64 lanes x 32 / 8 clk x 2.4 GHz = 629 T bit-MAC/s (microbenchmark of that mix: 588).  Round 4 counts
# MISMATCHES with the full-rate v_xor_b32 instead: 2 + 4 = 6 clk nominal per 32 bit-MACs and lane = 839 T bit-MAC/s.
# The two kinds do not overlap as the sum of their rates suggests: a kernel of nothing but 8 x v_xor then 8 x v_bcnt
# reaches 3.89 clk per instruction (32-long runs: 3.76) = 647 T bit-MAC/s -- the measured ceiling of this mix.
SPLIT_K_MAX_BATCH = 8               # images per GPU up to which the FP32 leg turns split K on (yl_network_set_split_k): the shards of a strong-scaled
                                    # 8-GPU run, whose 19 x 19 layers are one workgroup's K loop long (measured: +5.6 % at 8 images, +-0 at 16)
VALU_POPC_PEAK_TBITMAC = 839.0
VALU_POPC_MEASURED_TBITMAC = 647.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="yolov3", choices=["yolov3", "yolov3-tiny", "tiny-yolo-xnor"])
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch (strong scaling) / images per GPU (weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--mode", default="both", choices=["both", "fp32", "int8", "bf16"],
                    help="both = FP32 leg as `value` + INT8 leg under \"int8\"; fp32 / int8 = that leg only as `value`")
    ap.add_argument("--thresh", type=float, default=0.24)
    ap.add_argument("--cap", type=int, default=1024, help="detection records per image")
    ap.add_argument("--nms", type=float, default=0.4, help="do_nms_sort threshold (src/main.c:173); 0 = compaction only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-extras", action="store_true", help="skip batch sweep / group leg / agreement / recalibration")
    ap.add_argument("--raw-head", action="store_true",
                    help="keep the uncalibrated random detection head (thousands of boxes per image)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--layers", action="store_true", help="also print a per-layer table to stderr")
    ap.add_argument("--tile", type=int, default=0, help="force K1 tile config (tuning)")
    ap.add_argument("--no-winograd", action="store_true", help="A/B: 3x3 stride-1 layers on the direct kernels (yl_network_set_winograd 0)")
    ap.add_argument("--variant", type=int, default=-1, help="FP32 schedule variant bits (yl_network_set_variant; A/B runs)")
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


def calibrate_head(Network, cfg: str, wts: str, size: int, device: int, thresh: float):
    """Bias shifts that give the random-weight YOLO heads a detector-like output density.
    With i.i.d. weights whole anchor channels saturate (every cell of an anchor passes the
    objectness threshold with ~40 of 80 classes each: ~3800 boxes/image at 608), which no trained
    detector produces and which would turn the post-processing stage into the benchmark.  A
    2-image FP32 probe measures each head channel's logit distribution; the objectness channels are
    shifted so that 0.3 % of the cells pass `thresh` (~70 boxes per 608x608 image, COCO-like) and
    the class channels so that ~1.5 % of (box, class) pairs exceed 0.5.  Convolution work is
    unchanged (same shapes, same FLOPs); only head biases move.  The same weights serve both legs.
    YL_HEAD_CACHE=<dir>: the shifts are kept there per (cfg name, size, thresh) and reused, so that a profiled run
    (tools/gpu_round.sh stats4) contains no launches of this batch-2 probe."""
    cache = None
    if os.environ.get("YL_HEAD_CACHE"):
        os.makedirs(os.environ["YL_HEAD_CACHE"], exist_ok=True)
        cache = os.path.join(os.environ["YL_HEAD_CACHE"], "%s_%d_%g.npz" % (os.path.basename(cfg), size, thresh))
        if os.path.exists(cache):
            with np.load(cache) as z:
                return [z["d%d" % k] for k in range(len(z.files))]
    probe = Network.load(cfg, wts, 2, 0, device=device)
    x = np.random.default_rng(11).random((2, 3, size, size), dtype=np.float32)
    probe.predict(x)
    deltas = []
    for i in range(probe.n):
        li = probe.layer_info(i)
        if li["type"] not in (21, 22):        # YL_REGION, YL_YOLO: both take logistic(objectness) of channel 4 of an anchor
            continue
        per = 5 + li["classes"]
        o = probe.layer_output(i - 1).reshape(2, li["n"], per, -1)
        d = np.zeros((li["n"], per), dtype=np.float32)
        target_obj = float(np.log(thresh / (1.0 - thresh)))
        for a in range(li["n"]):
            d[a, 4] = target_obj - np.quantile(o[:, a, 4, :], 0.997)
            if li["type"] == 22:              # [region] classes go through a softmax: their logits stay as they are
                for c in range(5, per):
                    d[a, c] = 0.0 - np.quantile(o[:, a, c, :], 0.985)
        deltas.append(d.reshape(-1))
    probe.close()
    if cache:
        np.savez(cache, **{"d%d" % k: d for k, d in enumerate(deltas)})
    return deltas


def recalibrate_int8(Network, cfg: str, wts: str, size: int, device: int) -> str:
    """`darknet detector calibrate` on synthetic images (yl_network_calibrate): the cfg's shipped
    input_calibration= list belongs to the trained weights; the -quantized leg gets the multipliers the
    reference's own tool derives for THESE weights.  Returns the path of a cfg with that list."""
    cache = None
    mult = None
    if os.environ.get("YL_HEAD_CACHE"):           # as in calibrate_head: keep the probe out of profiled runs
        os.makedirs(os.environ["YL_HEAD_CACHE"], exist_ok=True)
        cache = os.path.join(os.environ["YL_HEAD_CACHE"], "%s_%d_int8mult.npy" % (os.path.basename(cfg), size))
        if os.path.exists(cache):
            mult = np.load(cache)
    if mult is None:
        probe = Network.load(cfg, wts, 2, 0, device=device)
        imgs = np.random.default_rng(12).random((4, 3, size, size), dtype=np.float32)
        mult = probe.calibrate(imgs)
        probe.close()
        if cache:
            np.save(cache, np.asarray(mult, dtype=np.float64))
    text = open(cfg).read()
    vals = ", ".join("%.6g" % float(v) for v in mult) + ", 16"
    lines = []
    done = False
    for line in text.splitlines():
        if line.strip().startswith("input_calibration"):
            line = "input_calibration = " + vals
            done = True
        lines.append(line)
    if not done:
        k = next(i for i, ln in enumerate(lines) if ln.strip().startswith("["))
        lines.insert(k + 1, "input_calibration = " + vals)
    out = cfg[:-4] + "-recal.cfg"
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    return out


def wino_executed_flops(li: dict, B: int) -> float:
    """MFMA FLOPs conv_f32_wino32 issues: 16 multiplies per (filter, channel, 2x2 tile), filters padded to the
    32-row tile, tiles = ceil(h/2)*ceil(w/2) per image padded to the 64-tile workgroup."""
    tiles = B * ((li["h"] + 1) // 2) * ((li["w"] + 1) // 2)
    tiles = (tiles + 63) // 64 * 64
    m = (li["n"] + 31) // 32 * 32
    return 2.0 * m * li["c"] * 16.0 * tiles


def row3_executed_flops(li: dict, B: int, name: str) -> float:
    """BF16-MFMA FLOPs conv_f32_row3 issues: four planes x six piece products per (filter, channel, filter row, tile of two
    output columns), filters padded to 32, tiles = B * h * ceil(w / 2) padded to the workgroup's tile count (in the name)"""
    import re
    m = re.search(r"<(\d+)x(\d+)t", name)
    bt = int(m.group(2)) if m else 128
    tiles = B * li["h"] * ((li["w"] + 1) // 2)
    tiles = (tiles + bt - 1) // bt * bt
    mm = (li["n"] + 31) // 32 * 32
    return 2.0 * mm * (3.0 * li["c"]) * 4.0 * tiles * 6.0


def kernel_pipe(name: str) -> str:
    """which execution pipe a convolution kernel instance issues its multiplies on"""
    if name.startswith("conv_f32_row3") or name.startswith("conv_f32_x3") or name.startswith("conv_bf16"):
        return "bf16_mfma"
    if name.startswith("conv_f32_first"):
        # K1m instances (conv_f32_first<mfma16x16x4,...> / <mfma32x32x2,...>) run on the FP32 matrix instruction, K1f on the VALU
        return "fp32_mfma" if "<mfma" in name else "fp32_valu"
    if name.startswith("conv_i8"):
        return "int8_mfma"
    if name.startswith("conv_xnor"):
        return "valu_popcount"
    return "fp32_mfma"


def iou(a, b):
    def ov(x1, w1, x2, w2):
        return min(x1 + w1 / 2, x2 + w2 / 2) - max(x1 - w1 / 2, x2 - w2 / 2)
    w = ov(a[0], a[2], b[0], b[2]); h = ov(a[1], a[3], b[1], b[3])
    if w <= 0 or h <= 0:
        return 0.0
    inter = w * h
    return inter / (a[2] * a[3] + b[2] * b[3] - inter)


def detection_agreement(ref_rows, got_rows):
    """`got` (INT8) against `ref` (FP32) per image: a reference detection (a row with a surviving class) is
    matched by the INT8 detection of the same best class with the highest IoU >= .5; greedy, one-to-one."""
    tp = n_ref = n_got = 0
    ious = []
    dobj = []
    for r, g in zip(ref_rows, got_rows):
        r = r[(r[:, 6:] > 0).any(axis=1)] if len(r) else r
        g = g[(g[:, 6:] > 0).any(axis=1)] if len(g) else g
        n_ref += len(r); n_got += len(g)
        used = np.zeros(len(g), bool)
        gc = g[:, 6:].argmax(axis=1) if len(g) else np.zeros(0, int)
        for row in r:
            c = int(row[6:].argmax())
            best, bj = 0.0, -1
            for j in np.flatnonzero((gc == c) & ~used):
                v = iou(row[:4], g[j, :4])
                if v > best:
                    best, bj = v, j
            if bj >= 0 and best >= 0.5:
                used[bj] = True
                tp += 1
                ious.append(best)
                dobj.append(abs(float(row[4]) - float(g[bj, 4])))
    return {
        "fp32_detections": int(n_ref), "int8_detections": int(n_got), "matched": int(tp),
        "recall_vs_fp32": tp / n_ref if n_ref else None, "precision_vs_fp32": tp / n_got if n_got else None,
        "mean_iou_of_matched": float(np.mean(ious)) if ious else None,
        "mean_abs_objectness_diff": float(np.mean(dobj)) if dobj else None,
        "rule": "same best class, IoU >= 0.5, one-to-one; thresh and nms as the timed step",
    }


class Leg:
    """One timed leg (FP32 or INT8) of the workload on this rank."""

    def __init__(self, args, torch, dist, dev, stream, Network, cfg, wts, quantized, b_local, world, use_dist):
        self.args, self.torch, self.dist = args, torch, dist
        self.quantized, self.B, self.world, self.use_dist = quantized, b_local, world, use_dist
        self.stream = stream
        # quantized: 0 FP32, 1 -quantized INT8, 2 the opt-in BF16 variant of the FP32 path
        self.net = Network.load(cfg, wts, b_local, 1 if quantized == 1 else 0, device=dev.index, fuse=not args.no_fuse,
                                bf16=(quantized == 2), variant=(args.variant if args.variant >= 0 else None),
                                winograd=not args.no_winograd, split_k=(quantized == 0 and b_local <= SPLIT_K_MAX_BATCH))
        self.net.set_stream(stream.cuda_stream)
        if args.tile:
            self.net.set_conv_tile(args.tile)
        if args.i8_tile:
            self.net.set_int8_tile(args.i8_tile)
        last = self.net.layer_info(self.net.n - 1)
        self.classes = last["classes"]
        self.rec = torch.zeros((b_local, args.cap, 6 + self.classes), device=dev, dtype=torch.float32)
        self.cnt = torch.zeros((b_local,), device=dev, dtype=torch.int32)
        if use_dist:
            # strong scaling shards may be uneven: every rank contributes max-shard rows, the tail is padding
            self.bmax = -(-args.global_batch // world)
            self.rec_pad = torch.zeros((self.bmax, args.cap, 6 + self.classes), device=dev, dtype=torch.float32)
            self.cnt_pad = torch.zeros((self.bmax,), device=dev, dtype=torch.int32)
            self.rec_all = torch.zeros((world * self.bmax, args.cap, 6 + self.classes), device=dev, dtype=torch.float32)
            self.cnt_all = torch.zeros((world * self.bmax,), device=dev, dtype=torch.int32)

    def step(self, x, slot):
        args, torch, net = self.args, self.torch, self.net
        with torch.cuda.stream(self.stream):
            net.forward_timed(x.data_ptr(), slot)   # HIP events around every layer, no host sync
            if args.nms > 0:    # get_network_boxes + do_nms_sort for every image, on the GPU
                net.detect_batch(args.thresh, args.nms, args.cap, self.rec.data_ptr(), self.cnt.data_ptr())
            else:
                net.compact_detections(args.thresh, args.cap, self.rec.data_ptr(), self.cnt.data_ptr())
            if self.use_dist:
                self.rec_pad[:self.B].copy_(self.rec, non_blocking=True)
                self.cnt_pad[:self.B].copy_(self.cnt, non_blocking=True)
                self.dist.all_gather_into_tensor(self.rec_all, self.rec_pad)
                self.dist.all_gather_into_tensor(self.cnt_all, self.cnt_pad)

    def run(self, x, steps, warmup):
        torch, dist = self.torch, self.dist
        MAX_SLOTS = 64      # HIP-event timing slots: one per timed step (wraps beyond 64 steps)
        for _ in range(warmup):
            self.step(x, 0)
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(x, k % MAX_SLOTS)
        torch.cuda.synchronize()
        if self.use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        n_slots = min(steps, MAX_SLOTS)
        layer_ms = np.zeros(self.net.n, dtype=np.float64)
        for sl in range(n_slots):
            ms, _ = self.net.layer_times(sl)
            layer_ms[:] += ms
        self.layer_ms = layer_ms / n_slots           # per step
        if self.use_dist:
            t = torch.tensor([elapsed], device=x.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def sample_sclk(self, x, device_index: int, samples: int = 3):
        """The shader clock the chip grants THIS workload: `rocm-smi --showclocks` a few times from a helper thread while
        untimed steps keep running (after the timed region: nothing here touches `value`).  Returns MHz (median) or None.
        With K1x on the direct layers the part runs the whole step at ~2.07 GHz / 1.17 kW instead of 2.35 GHz / 1.31 kW
        (profiles/r4_clock_power_x3_vs_fp32_mfma.txt): the peaks of MI355X_MICROARCH.md are quoted at 2.4 GHz."""
        import re
        import shutil
        import subprocess
        import threading
        smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        if not os.path.exists(smi):
            return None
        got, done = [], threading.Event()

        def work():
            try:
                for _ in range(samples):
                    r = subprocess.run([smi, "-d", str(device_index), "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                       timeout=20, text=True)
                    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", r.stdout)
                    if m:
                        got.append(int(m.group(1)))
            except Exception:
                pass
            done.set()

        for _ in range(3):
            self.step(x, 0)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        t_end = time.perf_counter() + 30.0
        while not done.is_set() and time.perf_counter() < t_end:
            self.step(x, 0)
            self.torch.cuda.synchronize()
        th.join(timeout=5)
        return float(np.median(got)) if got else None

    def kernels(self):
        """per kernel instance: launches, ms per step, algorithmic + executed flops, algorithmic bytes"""
        net, B = self.net, self.B
        kern = {}
        for i, li in enumerate(net.layers()):
            if li["type"] != 0:
                continue
            # the instances that also write the pooled tensor of a fused [maxpool] are the same kernel for the roofline
            name = net.layer_kernel(i).replace(",pool+", "").replace(",pool", "")
            flops = 2.0 * li["n"] * li["size"] ** 2 * li["c"] * li["out_h"] * li["out_w"] * B
            rd, wr = net.layer_traffic(i)
            k = kern.setdefault(name, {"flops": 0.0, "exec_flops": 0.0, "bytes": 0.0, "ms": 0.0, "launches": 0})
            k["flops"] += flops
            if "wino" in name:
                k["exec_flops"] += wino_executed_flops(li, B)          # FP32 MFMA: 16 multiplies per 2x2 tile
            elif name.startswith("conv_f32_row3"):
                k["exec_flops"] += row3_executed_flops(li, B, name)    # BF16 MFMA: 4 planes x 6 piece products per tile pair
            elif name.startswith("conv_f32_x3"):
                k["exec_flops"] += 6.0 * flops                         # BF16 MFMA: 6 piece products per multiply
            else:
                k["exec_flops"] += flops
            k["bytes"] += rd + wr
            k["ms"] += float(self.layer_ms[i])
            k["launches"] += 1
        return kern

    def post_ms(self, reps=5):
        torch = self.torch
        args = self.args
        with torch.cuda.stream(self.stream):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            for _ in range(reps):
                if args.nms > 0:
                    self.net.detect_batch(args.thresh, args.nms, args.cap, self.rec.data_ptr(), self.cnt.data_ptr())
                else:
                    self.net.compact_detections(args.thresh, args.cap, self.rec.data_ptr(), self.cnt.data_ptr())
            e1.record(self.stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def head_tensors(self, images=8):
        """YOLO/REGION layer outputs of the first few images (for the INT8-vs-FP32 agreement report)"""
        self.torch.cuda.synchronize()
        out = []
        for i, li in enumerate(self.net.layers()):
            if li["type"] in (21, 22):
                out.append(np.stack([self.net.layer_output_image(i, b) for b in range(min(images, self.B))]))
        return out

    def rows(self):
        """detections of this rank's images on the host (after a step)"""
        self.torch.cuda.synchronize()
        rows, _ = self.net.get_boxes_batch(self.args.thresh, self.args.nms if self.args.nms > 0 else 0.4, cap=self.args.cap)
        return rows

    def close(self):
        self.net.close()
        self.rec = self.cnt = None
        if self.use_dist:
            self.rec_pad = self.cnt_pad = self.rec_all = self.cnt_all = None
        self.torch.cuda.empty_cache()


def pmc_traffic(leg, kernel_name: str):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE in their own runs, gfx950 FETCH_SIZE x2 correction), collected on the batch-64 step of
    the same model and size; other per-GPU batches (strong-scaled ranks) are scaled linearly.  A missing entry is an
    ERROR (a renamed kernel must not silently lose its counter evidence): said loudly on stderr and in the line."""
    a = leg.args
    key = kernel_name if (a.model == "yolov3" and a.size == 608) else "%s@%s-%d" % (kernel_name, a.model, a.size)
    ref_batch = {"yolov3": 64, "yolov3-tiny": 32, "tiny-yolo-xnor": 128}.get(a.model, 64)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f).get(key)
    except (OSError, ValueError) as ex:
        pt = None
        print("bench.py: cannot read profiles/pmc_traffic.json: %r" % (ex,), file=sys.stderr)
    if not pt:
        print("bench.py: ERROR: no PMC traffic entry for kernel %r in profiles/pmc_traffic.json -- re-run the "
              "FETCH_SIZE / WRITE_SIZE passes (tools/gpu_round.sh pmc) for the shipped kernel" % key, file=sys.stderr)
        return None, "MISSING from profiles/pmc_traffic.json: %s" % key
    traffic = (2.0 * pt["fetch_kib"] + pt["write_kib"]) * 1024.0 * (leg.B / float(ref_batch))
    src = "committed rocprofv3 PMC passes (profiles/pmc_traffic.json, batch %d), not collected in this run" % ref_batch
    if leg.B != ref_batch:
        src += "; scaled linearly to this rank's batch of %d" % leg.B
    return traffic, src


_KERNEL_SYMBOL = {          # bench kernel instance -> substring of the C++ kernel name rocprofv3 reports
    "conv_f32_wino": "conv_f32_wino32_kernel<", "conv_f32_row3": "conv_f32_row3_kernel<", "conv_f32_x3": "conv_f32_x3_kernel<",
    "conv_i8_mfma<128x128>": "conv_i8_mfma_kernel<128, 128",
    "conv_bf16_mfma<128x128>": "conv_bf16_mfma_kernel<128, 128", "conv_xnor": "conv_xnor_kernel<",
}


def live_pmc_traffic(args, kernel_name: str, mode: str, timeout_s: float = 240.0):
    """HBM bytes per launch of `kernel_name`, MEASURED in this invocation: two child runs of this script (one step of the
    same workload, --no-extras --raw-head --nms 0 so that only launches of the bench batch exist) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- separate passes, no other tracing domain,
    as MI355X_MICROARCH.md prescribes -- mean over the kernel's dispatches, FETCH_SIZE x2 (gfx950 counts 64 B per 128-B
    request), counters are KiB.  Returns (bytes, source) or (None, reason); the committed passes (pmc_traffic.json)
    stay the fallback."""
    import csv
    import glob
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    sym = next((v for k, v in _KERNEL_SYMBOL.items() if kernel_name.startswith(k)), None)
    import re
    m = re.match(r"conv_f32_(row3|x3)<(\d+)x(\d+)", kernel_name)
    if m:           # the template instance of THIS tile (a step also launches other tiles of the same kernel on other layers)
        sym = "conv_f32_%s_kernel<%s, %s," % (m.group(1), m.group(2), m.group(3))
    if sym is None:
        return None, "no kernel symbol known for %s" % kernel_name
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="yl_pmc_%s_" % counter.lower())
        cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--model", args.model, "--size", str(args.size),
               "--batch", str(args.batch), "--mode", mode, "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
               "--no-e2e", "--no-extras", "--raw-head", "--nms", "0"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError) as ex:
            return None, "rocprofv3 pass %s failed: %r" % (counter, ex)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, "rocprofv3 pass %s: exit %d, %d csv" % (counter, r.returncode, len(files))
        tot, n = 0.0, 0
        for f in files:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if sym in (row.get("Kernel_Name") or "") and (row.get("Counter_Name") or "") == counter:
                        tot += float(row.get("Counter_Value") or 0.0)
                        n += 1
        shutil.rmtree(out, ignore_errors=True)
        if n == 0:
            return None, "no dispatch of %s in the %s pass" % (sym, counter)
        vals[counter] = (tot / n, n)
    traffic = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in two child runs of one "
                     "step of this workload (%d dispatches each), FETCH_SIZE %.1f KiB x2 + WRITE_SIZE %.1f KiB per launch"
                     % (vals["FETCH_SIZE"][1], vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]))


def torchrun_world1(args, timeout_s: float = 300.0):
    """The driver's N > 1 launch form at world size 1 on this box: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1` -- the one-process-per-GPU path (RCCL process group, all-gather of the detection records inside the
    step) executes on every default run, not only when a multi-GPU node is available."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", args.model, "--size", str(args.size),
           "--batch", str(args.batch), "--mode", "fp32", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-e2e", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        return {"error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
    d = json.loads(line)
    return {"value": d["value"], "unit": "images/sec", "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"],
            "what": "python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --mode fp32 --steps 5: RCCL process "
                    "group of one rank, all_gather_into_tensor of the detection records inside every step"}


def fp32_roofline(leg, args):
    """Roofline block of the FP32 leg.  Since round 4 the FP32 convolutions issue their multiplies on THREE pipes: the BF16
    matrix pipe (K1r row-wise Winograd and K1x direct layers: FP32 operands as exact sums of three bf16 pieces, six piece
    products per multiply), the FP32 matrix instruction (2-D Winograd with a folded [maxpool], layers with C % 16 != 0, <= 32
    filters) and the FP32 vector ALU (first layer).  `achieved` / `peak` / `frac` are the dominant kernel's on ITS pipe --
    executed (issued) FLOPs incl. tile padding / measured launch time; the algorithmic rate 2*M*K*N is separate."""
    peaks = {"bf16_mfma": BF16_MFMA_PEAK_TFLOPS, "fp32_mfma": FP32_MATRIX_PEAK_TFLOPS, "fp32_valu": FP32_VECTOR_PEAK_TFLOPS}
    kern = leg.kernels()
    dom_name = max(kern, key=lambda n: kern[n]["flops"])
    dom = kern[dom_name]
    pipe = kernel_pipe(dom_name)
    peak = peaks[pipe]
    sec = dom["ms"] * 1e-3
    executed = dom["exec_flops"] / sec / 1e12 if sec > 0 else 0.0
    algorithmic = dom["flops"] / sec / 1e12 if sec > 0 else 0.0
    conv_ms = sum(k["ms"] for k in kern.values())
    conv_flops = sum(k["flops"] for k in kern.values())
    traffic, traffic_src = pmc_traffic(leg, dom_name)
    n = max(dom["launches"], 1)
    per_pipe = {}
    for nm, k in kern.items():
        pp = per_pipe.setdefault(kernel_pipe(nm), {"ms_per_step": 0.0, "executed_flops": 0.0, "algorithmic_flops": 0.0, "launches": 0})
        pp["ms_per_step"] += k["ms"]; pp["executed_flops"] += k["exec_flops"]; pp["algorithmic_flops"] += k["flops"]
        pp["launches"] += k["launches"]
    for pn, pp in per_pipe.items():
        t = pp["ms_per_step"] * 1e-3
        pp["share_of_conv_time"] = pp["ms_per_step"] / conv_ms if conv_ms > 0 else 0.0
        pp["executed_tflops"] = pp.pop("executed_flops") / t / 1e12 if t > 0 else 0.0
        pp["algorithmic_tflops"] = pp.pop("algorithmic_flops") / t / 1e12 if t > 0 else 0.0
        pp["peak_tflops"] = peaks[pn]
        pp["frac"] = pp["executed_tflops"] / peaks[pn]
    all_alg = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    return {
        "bound": "mfma", "pipe": pipe, "kernel": dom_name,
        "achieved": executed, "peak": peak, "unit": "TFLOP/s",
        "frac": executed / peak,
        "achieved_is": {"bf16_mfma": "issued BF16-MFMA FLOPs (six bf16 piece products per FP32 multiply; row-wise Winograd: 4 planes per "
                                     "filter row and tile of two output columns; incl. tile padding) / measured launch time, vs the dense "
                                     "bf16 MFMA peak",
                        "fp32_mfma": "executed FP32-MFMA FLOPs (Winograd: 16 multiplies per 2x2 tile incl. tile padding) / measured launch time",
                        "fp32_valu": "fma FLOPs / measured launch time"}[pipe],
        "algorithmic_tflops": algorithmic,                      # 2*M*K*N per launch / the same time
        "algorithmic_vs_fp32_matrix_peak": algorithmic / FP32_MATRIX_PEAK_TFLOPS,
        "issued_per_algorithmic_flop": dom["exec_flops"] / dom["flops"] if dom["flops"] else None,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "algorithmic_bytes_per_launch": dom["bytes"] / n,
        "algorithmic_flops_per_launch": dom["flops"] / n, "executed_flops_per_launch": dom["exec_flops"] / n,
        "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / n,
        "all_conv_algorithmic_tflops": all_alg,
        "all_conv_algorithmic_vs_fp32_matrix_peak": all_alg / FP32_MATRIX_PEAK_TFLOPS,   # north_star's target: >= 0.70
        "per_pipe": per_pipe,
        "conv_ms_per_step": conv_ms, "other_layers_ms_per_step": float(leg.layer_ms.sum() - conv_ms),
        "by_kernel": {nm: {"pipe": kernel_pipe(nm),
                           "executed_tflops": (k["exec_flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0),
                           "frac_of_pipe_peak": (k["exec_flops"] / (k["ms"] * 1e-3) / 1e12 / peaks.get(kernel_pipe(nm), 1e30) if k["ms"] > 0 else 0.0),
                           "algorithmic_tflops": (k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0),
                           "algorithmic_gbs": (k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0),
                           "ms_per_step": k["ms"], "launches": k["launches"]} for nm, k in kern.items()},
    }


def int8_roofline(leg, prefix="conv_i8", mfma_peak=None, what="int8"):
    mfma_peak = mfma_peak or INT8_MFMA_PEAK_TOPS
    kern = leg.kernels()
    i8 = {n: k for n, k in kern.items() if n.startswith(prefix)}
    if not i8:
        return None
    dom_name = max(i8, key=lambda n: i8[n]["ms"])
    dom = i8[dom_name]
    sec = dom["ms"] * 1e-3
    n = max(dom["launches"], 1)
    gbs = dom["bytes"] / sec / 1e9 if sec > 0 else 0.0
    tops = dom["flops"] / sec / 1e12 if sec > 0 else 0.0
    tot_ms = sum(k["ms"] for k in i8.values())
    tot_bytes = sum(k["bytes"] for k in i8.values())
    tot_ops = sum(k["flops"] for k in i8.values())
    traffic, traffic_src = pmc_traffic(leg, dom_name)
    return {
        "bound": "hbm", "kernel": dom_name,
        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
        "achieved_is": "algorithmic bytes (%s in, weights, FP32 [shortcut] operand in / sum out, %s side output; "
                       "yl_network_layer_traffic) of the kernel's launches / their measured duration" % (what, what),
        "traffic": traffic,
        "traffic_source": traffic_src,
        "mfma_tops": tops, "mfma_peak_tops": mfma_peak, "mfma_frac": tops / mfma_peak,
        "algorithmic_bytes_per_launch": dom["bytes"] / n, "launches_per_step": dom["launches"],
        "avg_launch_ms": dom["ms"] / n,
        "all_int8_conv_gbs": tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0,
        "all_int8_conv_tops": tot_ops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
        "int8_conv_ms_per_step": tot_ms, "other_ms_per_step": float(leg.layer_ms.sum() - tot_ms),
        "by_kernel": {nm: {"algorithmic_gbs": (k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0),
                           "tops": (k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0),
                           "ms_per_step": k["ms"], "launches": k["launches"]} for nm, k in kern.items()},
    }


def xnor_roofline(leg):
    kern = leg.kernels()
    xn = {n: k for n, k in kern.items() if n.startswith("conv_xnor")}
    if not xn:
        return None
    # the template instances of conv_xnor_kernel (filter tile x word width x epilogue) are ONE kernel for the
    # roofline: the block below is over all of its launches of a step (7 for tiny-yolo-xnor)
    dom_name = "conv_xnor"
    dom = {f: sum(k[f] for k in xn.values()) for f in ("flops", "exec_flops", "bytes", "ms", "launches")}
    sec = dom["ms"] * 1e-3
    gbs = dom["bytes"] / sec / 1e9 if sec > 0 else 0.0
    tbm = dom["flops"] / 2 / sec / 1e12 if sec > 0 else 0.0
    traffic, traffic_src = pmc_traffic(leg, dom_name)
    # with sign words handed from layer to layer the bit convolutions move almost no HBM bytes: the roof that
    # binds them is the VALU's xnor + popcount rate (no 1-bit MFMA exists on CDNA4)
    return {
        "bound": "valu", "kernel": dom_name, "achieved": tbm, "peak": VALU_POPC_PEAK_TBITMAC, "unit": "Tbit-MAC/s",
        "frac": tbm / VALU_POPC_PEAK_TBITMAC,
        "measured_ceiling": VALU_POPC_MEASURED_TBITMAC, "frac_of_measured_ceiling": tbm / VALU_POPC_MEASURED_TBITMAC,
        "achieved_is": "9*C*M bit-MACs per output pixel of the XNOR convolutions / measured duration of their launches; "
                       "peak = one v_xor_b32 (full rate, 2 clk per wave64 instruction) + one accumulating v_bcnt_u32_b32 "
                       "(half rate, 4 clk) per 32 bit-MACs and lane, 1024 SIMDs, 2.4 GHz (rates measured: "
                       "profiles/r4_valu_issue_bench.txt); measured_ceiling = what a kernel of nothing but these two "
                       "instructions in the kernel's arrangement reaches on every CU (round 3's v_xnor form: peak 629, ceiling 588)",
        "hbm_gbs": gbs, "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS,
        "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
        "traffic": traffic, "traffic_source": traffic_src,
        "instances": {nm: {"launches": k["launches"], "ms_per_step": k["ms"],
                           "tbitmac_per_s": (k["flops"] / 2 / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0)}
                      for nm, k in xn.items()},
    }


def pcie_inclusive(net, torch, stream, args, B, rec, cnt, steps: int = 3):
    """Host-to-host rate of the same workload (reported next to `value`, never as `value`):
    (a) decoder-style input: B u8 768x576x3 frames in pageable host memory per step ->
        yl_network_set_input_u8 (pinned staging, H2D, GPU /255 + resize_image) -> forward ->
        batched detections + NMS on the GPU -> rows and counts to pinned host memory; steps are
        pipelined on one stream, one sync at the end;
    (b) the reference's own boundary: yl_network_predict(float CHW host images), which also
        brings the head tensors back (what network_predict_cpu leaves in l.output)."""
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 256, size=(576, 768, 3), dtype=np.uint8) for _ in range(8)]
    rec_h = torch.empty(rec.shape, dtype=rec.dtype).pin_memory()
    cnt_h = torch.empty(cnt.shape, dtype=cnt.dtype).pin_memory()

    batch_frames = [frames[b % len(frames)] for b in range(B)]

    def one(batched):
        if batched:
            net.set_input_u8_batch(batch_frames)          # ONE call: pool-copied, uploaded on the copy stream under the previous forward
        else:
            for b in range(B):
                net.set_input_u8(b, batch_frames[b])
        net.forward_staged()
        with torch.cuda.stream(stream):
            net.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr(),
                             sizes=(768, 576), relative=0)
            rec_h.copy_(rec, non_blocking=True)
            cnt_h.copy_(cnt, non_blocking=True)

    t_by = {}
    for batched in (False, True):
        one(batched)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(batched)
        torch.cuda.synchronize()
        t_by[batched] = (time.perf_counter() - t0) / steps
    t_u8 = t_by[True]
    x_host = np.random.default_rng(8).random((B, 3, args.size, args.size), dtype=np.float32)
    net.predict_raw(x_host)
    t0 = time.perf_counter()
    for _ in range(3):
        net.predict_raw(x_host)          # the C-ABI call itself (Network.predict would add a 377 MB numpy copy of the returned tensor)
    t_f32 = (time.perf_counter() - t0) / 3
    return {
        "value": B / t_u8, "unit": "images/sec", "ms_per_step": t_u8 * 1e3,
        "what": "u8 768x576x3 host frames -> yl_network_set_input_u8_batch -> forward -> yl_network_detect_batch -> "
                "rows+counts on the host; %d pipelined steps" % steps,
        "per_frame_calls": {"value": B / t_by[False], "ms_per_step": t_by[False] * 1e3,
                            "what": "the same with one yl_network_set_input_u8 per frame (round 5's form)"},
        "predict_float_host": {"value": B / t_f32, "unit": "images/sec", "ms_per_step": t_f32 * 1e3,
                               "what": "yl_network_predict(float CHW host batch): pinned staging + H2D of %.0f MB + forward + D2H of "
                                       "the heads, as a two-sub-batch pipeline on three streams" % (x_host.nbytes / 1e6)},
    }


def decode_inclusive(net, torch, stream, args, B, rec, cnt, threads=(1, 8, 32, 64)):
    """SURVEY 8f-2's open question, answered with a number: does HOST JPEG decode keep up with the GPU?
    The reference decodes with stb on one core per image (load_image_stb, src/additionally.c:3084-3106, called from
    src/main.c:187).  A 768x576 JPEG (the size of bin/dog.jpg; made here from a smooth synthetic frame, quality 90, because the
    reference tree does not travel to the GPU box) is decoded on N host threads (Pillow = libjpeg-turbo, the GIL is released
    inside the decoder) and every decoded frame goes through the same boundary the PCIe-inclusive leg uses
    (yl_network_set_input_u8_batch -> GPU /255 + resize_image -> forward -> detect_batch -> rows on the host).
    Reported: decode-only img/s and decode-inclusive img/s per thread count, and the reference's own front end
    (load_image + resize_image on one core, oracle/_ref) beside them."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    try:
        from PIL import Image
    except Exception as ex:
        return {"error": "Pillow not importable: %r" % (ex,)}
    yy, xx = np.mgrid[0:576, 0:768].astype(np.float32)
    rng = np.random.default_rng(11)
    frame = np.stack([127 + 90 * np.sin(xx / 37.0 + c) * np.cos(yy / 23.0 - c) + rng.normal(0, 6, xx.shape) for c in range(3)],
                     axis=-1).clip(0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(frame).save(buf, format="JPEG", quality=90)
    jpeg = buf.getvalue()

    def decode(_):
        im = Image.open(io.BytesIO(jpeg))
        im.draft("RGB", im.size)
        return np.asarray(im.convert("RGB"))

    rec_h = torch.empty(rec.shape, dtype=rec.dtype).pin_memory()
    cnt_h = torch.empty(cnt.shape, dtype=cnt.dtype).pin_memory()
    out = {"jpeg_bytes": len(jpeg), "frame": "768x576x3", "decoder": "Pillow %s (libjpeg-turbo)" % getattr(Image, "__version__", "?"),
           "host_cores": os.cpu_count(), "by_threads": {}}
    for nt in threads:
        if nt > (os.cpu_count() or 1):
            continue
        with ThreadPoolExecutor(max_workers=nt) as ex:
            list(ex.map(decode, range(min(B, 2 * nt))))         # warm the pool
            t0 = time.perf_counter()
            frames = list(ex.map(decode, range(B)))
            t_dec = time.perf_counter() - t0
            steps = 2
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net.set_input_u8_batch(list(ex.map(decode, range(B))))
                net.forward_staged()
                with torch.cuda.stream(stream):
                    net.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr(),
                                     sizes=(768, 576), relative=0)
                    rec_h.copy_(rec, non_blocking=True)
                    cnt_h.copy_(cnt, non_blocking=True)
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) / steps
        del frames
        out["by_threads"][str(nt)] = {"decode_only_images_per_sec": B / t_dec, "decode_inclusive_images_per_sec": B / t_all,
                                      "ms_per_step": t_all * 1e3}
    best = max(out["by_threads"].values(), key=lambda v: v["decode_inclusive_images_per_sec"]) if out["by_threads"] else None
    if best:
        out["value"] = best["decode_inclusive_images_per_sec"]
        out["unit"] = "images/sec"
        out["ms_per_step"] = best["ms_per_step"]
    try:                      # the reference's own front end on one core (stb decode + /255 + resize_image), via oracle/_ref
        from oracle import refbind
        if refbind.available():
            import ctypes as C
            lib = refbind._bind(refbind.GOLD)
            if lib is not None:
                tmp = os.path.join(tempfile.mkdtemp(prefix="yl_jpeg_"), "frame.jpg")
                with open(tmp, "wb") as f:
                    f.write(jpeg)
                dst = np.zeros(3 * args.size * args.size, dtype=np.float32)
                sw, sh = C.c_int(0), C.c_int(0)
                fp = dst.ctypes.data_as(C.POINTER(C.c_float))
                lib.ref_load_resized(tmp.encode(), args.size, args.size, fp, C.byref(sw), C.byref(sh))
                t0 = time.perf_counter()
                k = 8
                for _ in range(k):
                    lib.ref_load_resized(tmp.encode(), args.size, args.size, fp, C.byref(sw), C.byref(sh))
                out["reference_front_end_one_core_images_per_sec"] = k / (time.perf_counter() - t0)
    except Exception as ex:
        out["reference_front_end_error"] = repr(ex)
    out["what"] = ("JPEG decode on N host threads -> yl_network_set_input_u8_batch -> forward -> yl_network_detect_batch -> rows on the host; "
                   "`value` = the best thread count")
    return out


def strict_leg(args, torch, dev, stream, Network, cfg, wts, x):
    """YL_PRECISION_FP32_STRICT (every FP32 convolution on the direct FP32-matrix-instruction kernel): the same step, its img/s"""
    B = x.shape[0]
    net = Network.load(cfg, wts, B, 0, device=dev.index, fuse=not args.no_fuse, strict=True)
    net.set_stream(stream.cuda_stream)
    classes = net.layer_info(net.n - 1)["classes"]
    rec = torch.zeros((B, args.cap, 6 + classes), device=dev, dtype=torch.float32)
    cnt = torch.zeros((B,), device=dev, dtype=torch.int32)

    def one():
        with torch.cuda.stream(stream):
            net.forward_device(x.data_ptr())
            net.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr())
    one()
    torch.cuda.synchronize()
    steps = max(2, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    kernels = sorted({net.layer_kernel(i).split("<")[0] for i in range(net.n) if net.layer_kernel(i)})
    net.close()
    return {"value": B / t, "unit": "images/sec", "ms_per_step": t * 1e3, "steps": steps, "kernels": kernels,
            "what": "yl_network_set_precision(YL_PRECISION_FP32_STRICT): FP32-MFMA direct kernels only (no Winograd, no three-piece bf16)"}


def batch_sweep(args, torch, dev, stream, Network, cfg, wts, x, batches=(8, 16, 32), steps=4):
    """FP32 step rate of one GPU at the per-rank batch of a strong-scaled 8/4/2-GPU run"""
    out = {}
    for b in batches:
        if b >= x.shape[0]:
            continue
        # split K (yl_network_set_split_k) where a rank's shard leaves CUs idle: what a strong-scaled rank runs (Leg: b_local <= 16)
        net = Network.load(cfg, wts, b, 0, device=dev.index, fuse=not args.no_fuse, split_k=b <= SPLIT_K_MAX_BATCH)
        net.set_stream(stream.cuda_stream)
        classes = net.layer_info(net.n - 1)["classes"]
        rec = torch.zeros((b, args.cap, 6 + classes), device=dev, dtype=torch.float32)
        cnt = torch.zeros((b,), device=dev, dtype=torch.int32)
        xb = x[:b].contiguous()

        def one():
            with torch.cuda.stream(stream):
                net.forward_device(xb.data_ptr())
                net.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr())
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps
        out[str(b)] = {"images_per_sec": b / t, "ms_per_step": t * 1e3, "split_k": b <= SPLIT_K_MAX_BATCH,
                       "split_layers": sum(",split" in net.layer_kernel(i) for i in range(net.n)),
                       "predicted_node_images_per_sec_at_%d_gpus" % (x.shape[0] // b): (x.shape[0] // b) * b / t}
        net.close()
    return out


def group_leg(args, torch, dev, Network, cfg, wts, x, steps=4):
    """the step through yl_group_* (single process, RCCL send/recv gather) on this process's GPU"""
    from yolo2_light_amd import parallel
    B = x.shape[0]
    model = Network.load(cfg, wts, B, 0, fuse=not args.no_fuse)
    grp = parallel.Group(model, [dev.index])
    model.close()
    rec = torch.zeros((B, args.cap, 6 + grp.classes), device=dev, dtype=torch.float32)
    cnt = torch.zeros((B,), device=dev, dtype=torch.int32)

    def one():
        grp.forward([x.data_ptr()])
        grp.detect_batch(args.thresh, args.nms if args.nms > 0 else 0.4, args.cap, rec.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    one()
    grp.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    grp.synchronize()
    t = (time.perf_counter() - t0) / steps
    dets = int(cnt.sum().item())
    grp.close()
    return {"value": B / t, "unit": "images/sec", "ms_per_step": t * 1e3, "devices": 1, "detections": dets,
            "what": "yl_group_forward + yl_group_detect_batch (ncclSend/ncclRecv gather to the root) on 1 device"}


def side_leg(args, torch, dist, dev, stream, Network, weights, zoo, model, size, batch):
    """BASELINE configs 2 and 5 as short extra legs of the default line: the same step
    (forward + on-device decode + NMS, inputs resident in HBM) on yolov3-tiny 416 batch 32 FP32 / tiny-yolo-xnor 416
    batch 128, each with the roofline block of its dominant kernel."""
    import copy
    a = copy.copy(args)
    a.model, a.size, a.batch, a.global_batch = model, size, batch, batch
    steps, warmup = args.steps, args.warmup          # side legs honour --steps / --warmup
    work = tempfile.mkdtemp(prefix="yl_bench_side_")
    cfg = zoo.write_cfg(model, work, size, size)
    wts = os.path.join(work, "synthetic.weights")
    with open(cfg) as f:
        cfg_text = f.read()
    weights.write_synthetic_weights(cfg_text, wts, seed=1)
    if not args.raw_head:
        deltas = calibrate_head(Network, cfg, wts, size, dev.index, args.thresh)
        if deltas:
            weights.write_synthetic_weights(cfg_text, wts, seed=1, head_bias_delta=deltas)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2222222)
    x = torch.rand((batch, 3, size, size), generator=gen, device=dev, dtype=torch.float32)
    leg = Leg(a, torch, dist, dev, stream, Network, cfg, wts, 0, batch, 1, False)
    elapsed = leg.run(x, steps, warmup)
    xnor = model == "tiny-yolo-xnor"
    det_counts = leg.cnt.cpu().numpy()
    out = {
        "workload": "%s.cfg %dx%d batch=%d %s, 1 GPU, same step as `value`" % (model, size, size, batch,
                                                                              "BIT1-XNOR" if xnor else "FP32"),
        "value": batch * steps / elapsed, "unit": "images/sec", "ms_per_step": elapsed / steps * 1e3,
        "steps": steps, "warmup": warmup, "dtype": "u1" if xnor else "f32",
        "roofline": xnor_roofline(leg) if xnor else fp32_roofline(leg, a),
        "detections_per_image": {"mean": float(det_counts.mean()), "max": int(det_counts.max())},
        "gflop_per_image": leg.net.flops_per_image / 1e9,
    }
    leg.close()
    del x
    torch.cuda.empty_cache()
    return out


def int8_vs_reference_int8(Network, cfg_q, wts, size, device, thresh, nms, images=2):
    """Config 4's accuracy figure against the right baseline: the HIP -quantized path against
    the REFERENCE's own network_predict_quantized (src/yolov2_forward_network_quantized.c:1160, the library bench.py
    already times as cpu_baseline) on the same images -- head tensors and detections.  The integer arithmetic of
    the two is bit-exact layer by layer (tests/test_gpu_headline.py, teacher-forced); end to end the FP32 layers of the
    -quantized path (layer 0, the linear head convolutions) sum in a different order, and a last-bit difference
    there can flip an int8 code downstream, so the comparison is statistical."""
    from oracle import refbind
    if not refbind.available(fast=True):
        return None
    ref = refbind.RefNetwork(cfg_q, wts, 1, 1, fast=True)
    ref_fp = refbind.RefNetwork(cfg_q, wts, 1, 0, fast=True)     # the reference's FP32 path on the same images: how far ITS INT8 is from ITS FP32
    hip = Network.load(cfg_q, wts, 1, 1, device=device, fuse=True)
    rng = np.random.default_rng(2222222)
    heads = [i for i in range(hip.n) if hip.layer_info(i)["type"] in (21, 22)]
    ref_rows, hip_rows, ref_fp_rows = [], [], []
    fp_corr = {i: [] for i in heads}
    err2 = {i: 0.0 for i in heads}; nrm2 = {i: 0.0 for i in heads}; exact = {i: 0 for i in heads}; total = {i: 0 for i in heads}
    for _ in range(images):
        x = rng.random((1, 3, size, size), dtype=np.float32)
        ref.predict(x)
        hip.predict(x)
        for i in heads:
            r = ref.layer_output(i).astype(np.float64); g = hip.layer_output(i).astype(np.float64)
            err2[i] += float(((g - r) ** 2).sum()); nrm2[i] += float((r ** 2).sum())
            exact[i] += int((g == r).sum()); total[i] += r.size
        ref_rows.append(ref.get_detections(0, size, size, thresh, nms=nms))
        hip_rows.append(hip.get_boxes(0, size, size, thresh, nms=nms))
        q_heads = {i: ref.layer_output(i).astype(np.float64).ravel() for i in heads}
        ref_fp.predict(x)
        ref_fp_rows.append(ref_fp.get_detections(0, size, size, thresh, nms=nms))
        for i in heads:
            fp_corr[i].append(float(np.corrcoef(ref_fp.layer_output(i).astype(np.float64).ravel(), q_heads[i])[0, 1]))
    hip.close()
    agree = detection_agreement(ref_rows, hip_rows)
    ref_q_vs_fp = detection_agreement(ref_fp_rows, ref_rows)
    return {
        "reference_int8_vs_reference_fp32": {
            "what": "the REFERENCE's own network_predict_quantized against its own network_predict_cpu on the same images and "
                    "synthetic weights: what `agreement_vs_fp32` of the HIP legs has to be read against -- the reference's 8-bit scheme "
                    "itself does not reproduce FP32 detections on i.i.d. weights, and the HIP INT8 path reproduces the reference's INT8",
            "reference_fp32_detections": ref_q_vs_fp["fp32_detections"], "reference_int8_detections": ref_q_vs_fp["int8_detections"],
            "matched": ref_q_vs_fp["matched"], "recall": ref_q_vs_fp["recall_vs_fp32"], "precision": ref_q_vs_fp["precision_vs_fp32"],
            "head_pearson": [float(np.mean(fp_corr[i])) for i in heads]},
        "images": images,
        "reference_int8_detections": agree["fp32_detections"], "hip_int8_detections": agree["int8_detections"],
        "matched": agree["matched"], "recall_vs_reference_int8": agree["recall_vs_fp32"],
        "precision_vs_reference_int8": agree["precision_vs_fp32"], "mean_iou_of_matched": agree["mean_iou_of_matched"],
        "mean_abs_objectness_diff": agree["mean_abs_objectness_diff"], "rule": agree["rule"],
        "head_tensors": [{"layer": i, "rel_rms_err": (err2[i] / max(nrm2[i], 1e-300)) ** 0.5,
                          "bit_equal_fraction": exact[i] / max(total[i], 1)} for i in heads],
        "what": "HIP -quantized path vs the reference's network_predict_quantized (AVX=1 OPENMP=1 build) on the same "
                "%d synthetic 608-style images, batch 1 each" % images,
    }


# ---------------------------------------------------------------------------------------------------------------
# The result line.  The driver reads the LAST stdout line and it has to stay small (round 5's 20.7 KB line was not
# parsed): stdout gets compact_line(out) (< 4 KB, numbers only, no prose); everything else -- per-kernel / per-pipe
# dictionaries, side legs, agreement blocks, the prose that says what each number is -- goes to bench_detail.json
# (repo root, and gpurun_out/ when that exists) and a short summary on stderr.
LINE_LIMIT_BYTES = 4096


def _r(v, sig=5):
    """Round floats to `sig` significant digits (ints, strings, None, bools pass through)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    return v


def _pick(src, keys, sig=5):
    if not isinstance(src, dict):
        return None
    return {k: _r(src[k], sig) for k in keys if k in src and not isinstance(src[k], (dict, list))}


_ROOFLINE_KEYS = ("bound", "pipe", "kernel", "achieved", "peak", "unit", "frac", "frac_at_sclk", "sclk_mhz",
                  "algorithmic_tflops", "algorithmic_vs_fp32_matrix_peak", "traffic", "algorithmic_bytes_per_launch",
                  "launches_per_step", "avg_launch_ms", "mfma_frac", "mfma_tops", "measured_ceiling",
                  "frac_of_measured_ceiling")


def compact_line(out: dict) -> dict:
    """The driver's line: contract fields + roofline + cpu_baseline + one number per side leg.  No prose beyond
    config.workload; every dropped field is in bench_detail.json."""
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                        "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:300], "global_batch": cfg.get("global_batch"),
                      "parallelism": cfg.get("parallelism"), "gflop_per_image": _r(cfg.get("gflop_per_image"))}
    line["roofline"] = _pick(out.get("roofline"), _ROOFLINE_KEYS)
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict) and "value" in cpu:
        c = _pick(cpu, ("value", "unit", "cores", "kind"))
        c["sample"] = str(cpu.get("sample", ""))[:140]
        if isinstance(cpu.get("int8"), dict):
            c["int8_value"] = _r(cpu["int8"].get("value"))
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = cpu if cpu is None else {"error": str(cpu.get("error", cpu))[:120]}
    for leg in ("int8", "bf16"):
        if isinstance(out.get(leg), dict):
            d = _pick(out[leg], ("value", "unit", "ms_per_step", "speedup_vs_fp32", "detect_ms_per_step"))
            d["roofline"] = _pick(out[leg].get("roofline"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "mfma_frac",
                                                            "mfma_tops", "launches_per_step", "avg_launch_ms"))
            line[leg] = d
    line["detect_ms_per_step"] = _r(out.get("detect_ms_per_step"))
    pc = out.get("pcie_inclusive")
    if isinstance(pc, dict):
        line["pcie_inclusive"] = _pick(pc, ("value", "ms_per_step"))
        if isinstance(pc.get("predict_float_host"), dict):
            line["pcie_inclusive"]["predict_float_host"] = _r(pc["predict_float_host"].get("value"))
    bs = out.get("batch_sweep")
    if isinstance(bs, dict):
        line["batch_sweep"] = {k: _r(v.get("images_per_sec", v.get("value"))) if isinstance(v, dict) else None
                               for k, v in bs.items() if k != "error"}
    for key in ("group_n1", "torchrun_world1", "weak_scaling", "config2_yolov3_tiny_416_b32_fp32",
                "config5_tiny_yolo_xnor_416_b128", "decode_inclusive", "strict_fp32"):
        v = out.get(key)
        if isinstance(v, dict):
            d = _pick(v, ("value", "ms_per_step", "error"))
            if isinstance(v.get("roofline"), dict):
                d["frac"] = _r(v["roofline"].get("frac"))
            line[key] = d
    line["detail"] = "bench_detail.json"
    return line


def emit(out: dict):
    """Detail to bench_detail.json (+ gpurun_out/) and stderr; the compact line as the LAST stdout line."""
    line = compact_line(out)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT_BYTES:        # never let the line outgrow the driver again: drop side legs, keep the contract
        for k in ("config5_tiny_yolo_xnor_416_b128", "config2_yolov3_tiny_416_b32_fp32", "batch_sweep", "bf16",
                  "group_n1", "torchrun_world1", "pcie_inclusive", "weak_scaling"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT_BYTES:
                break
    detail = json.dumps(out, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
        except OSError as ex:
            print("bench.py: could not write %s/bench_detail.json: %r" % (d, ex), file=sys.stderr)
    print("bench.py detail (also in bench_detail.json):", file=sys.stderr)
    print(json.dumps(out), file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(text, flush=True)


def relaunch_one_rank_per_gpu(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start N ranks of this script
    (one process per GPU, RCCL) on THIS node and hand their exit code back.  Refuses loudly when the node
    has fewer than N devices instead of timing fewer GPUs than the line would then claim."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    print("bench.py: --gpus %d outside torch.distributed.run -> launching %d ranks (one per GPU, RCCL) on this node; "
          "%d device(s) visible" % (n, n, have), file=sys.stderr)
    if have < n and not os.environ.get("YL_BENCH_FORCE_LAUNCH"):     # the override exists for tests/test_bench_launch.py
        print("bench.py: --gpus %d needs %d visible GPUs, this node has %d: refusing to time fewer GPUs than asked for"
              % (n, n, have), file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", YL_BENCH_LAUNCH="self-relaunch")
    return subprocess.call(cmd, cwd=ROOT, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(relaunch_one_rank_per_gpu(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the line's n_gpus must be the number of ranks that ran"
                         % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set) -> always use the RCCL path, even at world 1,
    # so the single-GPU box exercises exactly the code the multi-GPU runs execute
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py rank %d of %d needs GPU %d: the HIP path has no CPU fallback (%d device(s) visible)"
                         % (rank, world, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from yolo2_light_amd import Network, parallel, weights, zoo

    if args.scaling == "strong":
        args.global_batch = args.batch
        lo, hi = parallel.shard_range(args.global_batch, rank, world)
        b_local = hi - lo
        if b_local < 1:
            raise SystemExit("global batch %d is smaller than the number of GPUs" % args.batch)
    else:
        args.global_batch = args.batch * world
        lo, b_local = rank * args.batch, args.batch

    work = tempfile.mkdtemp(prefix="yl_bench_r%d_" % rank)
    cfg = zoo.write_cfg(args.model, work, args.size, args.size)
    wts = os.path.join(work, "synthetic.weights")
    with open(cfg) as f:
        cfg_text = f.read()
    weights.write_synthetic_weights(cfg_text, wts, seed=1)
    if not args.raw_head:
        deltas = calibrate_head(Network, cfg, wts, args.size, local_rank, args.thresh)
        if deltas:
            weights.write_synthetic_weights(cfg_text, wts, seed=1, head_bias_delta=deltas)
    xnor_model = args.model == "tiny-yolo-xnor"
    do_fp32 = args.mode in ("both", "fp32")
    do_int8 = args.mode in ("both", "int8") and not xnor_model
    # the opt-in BF16 variant: its own mode, or an extra leg of the default line (reported under "bf16", never `value`)
    do_bf16 = (args.mode == "bf16" or (args.mode == "both" and not args.no_extras and world == 1)) and not xnor_model
    cfg_q = cfg
    if do_int8 and not args.no_extras:
        try:
            cfg_q = recalibrate_int8(Network, cfg, wts, args.size, local_rank)
        except Exception as ex:          # the shipped list still runs
            print("recalibration failed: %r" % (ex,), file=sys.stderr)

    # one explicit (non-default) HIP stream shared by our kernels and torch/RCCL so the
    # compaction -> all-gather dependency is ordinary stream order
    stream = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev)
    if args.scaling == "strong":
        gen.manual_seed(2222222)         # the same global image set for every world size
        x_all = torch.rand((args.global_batch, 3, args.size, args.size), generator=gen, device=dev, dtype=torch.float32)
        x = x_all[lo:lo + b_local].contiguous() if world > 1 else x_all
        del x_all
    else:
        gen.manual_seed(2222222 + rank)
        x = torch.rand((b_local, 3, args.size, args.size), generator=gen, device=dev, dtype=torch.float32)
    torch.cuda.empty_cache()

    result = {}
    rows_fp32 = heads_fp32 = None
    for quantized in ([0] if do_fp32 else []) + ([1] if do_int8 else []) + ([2] if do_bf16 else []):
        leg = Leg(args, torch, dist, dev, stream, Network, cfg_q if quantized == 1 else cfg, wts, quantized, b_local,
                  world, use_dist)
        elapsed = leg.run(x, args.steps, args.warmup)
        info = {"value": args.global_batch * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3}
        if rank == 0:
            det_counts = leg.cnt.cpu().numpy()
            info["detections_per_image"] = {"mean": float(det_counts.mean()), "max": int(det_counts.max())}
            info["detect_ms_per_step"] = leg.post_ms()
            if xnor_model:
                info["roofline"] = xnor_roofline(leg)
            elif quantized == 2:
                info["roofline"] = int8_roofline(leg, "conv_bf16", BF16_MFMA_PEAK_TFLOPS, "bf16")
            elif quantized:
                info["roofline"] = int8_roofline(leg)
            else:
                info["roofline"] = fp32_roofline(leg, args)
                fam = {}
                for nm, k in leg.kernels().items():
                    f = nm.split("<")[0]
                    fam[f] = fam.get(f, 0) + k["launches"]
                info["arithmetic"] = (
                    "FP32 tensors in and out of every layer, FP32 accumulation; of the %d convolutions per step %d run as row-wise Winograd "
                    "F(2,3) and %d as direct convolutions on the BF16 MFMA pipe with every operand the exact sum of three bf16 pieces "
                    "(six piece products per multiply, FP32-class accuracy: tests/test_gpu_parity.py::test_fp32_error_vs_float64_truth), "
                    "%d as Winograd F(2x2,3x3) and %d as direct convolutions on the FP32 MFMA, %d on the FP32 vector ALU" % (
                        sum(fam.values()), fam.get("conv_f32_row3", 0), fam.get("conv_f32_x3", 0), fam.get("conv_f32_wino", 0),
                        fam.get("conv_f32_mfma_pipe", 0) + fam.get("conv_f32_smallk", 0), fam.get("conv_f32_first", 0)))
                if world == 1 and not args.no_extras:
                    clk = leg.sample_sclk(x, dev.index)
                    if clk:
                        rl = info["roofline"]
                        rl["sclk_mhz"] = clk
                        rl["peak_at_sclk"] = rl["peak"] * clk / NOMINAL_SCLK_MHZ
                        rl["frac_at_sclk"] = rl["achieved"] / rl["peak_at_sclk"] if rl.get("achieved") else None
                        rl["sclk_note"] = ("shader clock reported by rocm-smi while this leg runs (median of 3 samples over untimed steps "
                                           "after the timed region); `peak` and `frac` are at the 2.4 GHz of MI355X_MICROARCH.md, "
                                           "`peak_at_sclk` / `frac_at_sclk` at the clock the part actually grants this workload")
            info["gflop_per_image"] = leg.net.flops_per_image / 1e9
            if args.layers:
                for i, li in enumerate(leg.net.layers()):
                    print("%3d type=%2d %-28s %8.3f ms" % (i, li["type"], leg.net.layer_kernel(i), leg.layer_ms[i]),
                          file=sys.stderr)
            if world == 1 and not args.no_extras:
                try:
                    rows = leg.rows()
                    heads = leg.head_tensors()
                    if quantized and rows_fp32 is not None:
                        info["agreement_vs_fp32"] = detection_agreement(rows_fp32, rows)
                        corr = []
                        for a, b in zip(heads_fp32, heads):
                            a64, b64 = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
                            corr.append({"pearson": float(np.corrcoef(a64, b64)[0, 1]),
                                         "rel_rms_err": float(np.sqrt(np.mean((a64 - b64) ** 2)) / max(np.sqrt(np.mean(a64 ** 2)), 1e-30))})
                        info["agreement_vs_fp32"]["head_tensors_first_8_images"] = corr
                        info["agreement_vs_fp32"]["note"] = (
                            "synthetic i.i.d. weights: quantisation noise is not damped the way a trained detector damps "
                            "it; the number describes this workload, the kernels are bit-exact against the reference's "
                            "-quantized CPU path (tests/test_gpu_headline.py)")
                        if quantized == 1:
                            info["agreement_vs_fp32"]["input_calibration"] = (
                                "recomputed for the synthetic weights with yl_network_calibrate (4 synthetic images)"
                                if cfg_q != cfg else "the cfg's shipped list")
                        else:
                            info["agreement_vs_fp32"]["note"] = (
                                "opt-in BF16 operands (nearest even), FP32 accumulation: outside the FP32 path's 1e-4 contract")
                    elif not quantized:
                        rows_fp32, heads_fp32 = rows, heads
                except Exception as ex:
                    info["agreement_vs_fp32"] = {"error": repr(ex)}
            if not quantized and world == 1 and not args.no_e2e:
                try:
                    info["pcie_inclusive"] = pcie_inclusive(leg.net, torch, stream, args, b_local, leg.rec, leg.cnt)
                except Exception as ex:      # the headline number must not depend on this leg
                    info["pcie_inclusive"] = {"error": repr(ex)}
                if not args.no_extras:
                    try:
                        info["decode_inclusive"] = decode_inclusive(leg.net, torch, stream, args, b_local, leg.rec, leg.cnt)
                    except Exception as ex:
                        info["decode_inclusive"] = {"error": repr(ex)}
        result[("fp32", "int8", "bf16")[quantized]] = info
        leg.close()

    weak = None
    if world > 1 and args.scaling == "strong" and do_fp32 and not args.no_extras:
        # the same model at `--batch` images PER GPU (weak scaling) beside the strong-scaled `value`: separates what the
        # sharding + RCCL gather cost from what the kernels lose on 64/N images per GPU
        import copy
        wargs = copy.copy(args)
        wargs.global_batch = args.batch * world
        del x
        torch.cuda.empty_cache()
        gen.manual_seed(2222222 + rank)
        xw = torch.rand((args.batch, 3, args.size, args.size), generator=gen, device=dev, dtype=torch.float32)
        leg = Leg(wargs, torch, dist, dev, stream, Network, cfg, wts, 0, args.batch, world, use_dist)
        el = leg.run(xw, args.steps, args.warmup)
        weak = {"value": wargs.global_batch * args.steps / el, "unit": "images/sec", "ms_per_step": el / args.steps * 1e3,
                "global_batch": wargs.global_batch, "images_per_gpu": args.batch, "scaling": "weak",
                "what": "FP32 leg with --batch images on EVERY GPU (global batch x%d), same step incl. the RCCL all-gather" % world}
        leg.close()
        x = xw

    if rank == 0:
        extras = {}
        if weak is not None:
            extras["weak_scaling"] = weak
        if world == 1 and not args.no_extras and do_fp32 and not xnor_model:
            try:
                extras["batch_sweep"] = batch_sweep(args, torch, dev, stream, Network, cfg, wts, x)
            except Exception as ex:
                extras["batch_sweep"] = {"error": repr(ex)}
            try:
                extras["group_n1"] = group_leg(args, torch, dev, Network, cfg, wts, x)
            except Exception as ex:
                extras["group_n1"] = {"error": repr(ex)}
            try:
                extras["strict_fp32"] = strict_leg(args, torch, dev, stream, Network, cfg, wts, x)
            except Exception as ex:
                extras["strict_fp32"] = {"error": repr(ex)}
            if args.model == "yolov3" and args.size == 608:
                del x
                torch.cuda.empty_cache()
                for key, (m, sz, b) in (("config2_yolov3_tiny_416_b32_fp32", ("yolov3-tiny", 416, 32)),
                                        ("config5_tiny_yolo_xnor_416_b128", ("tiny-yolo-xnor", 416, 128))):
                    try:
                        extras[key] = side_leg(args, torch, dist, dev, stream, Network, weights, zoo, m, sz, b)
                    except Exception as ex:
                        extras[key] = {"error": repr(ex)}
        if world == 1 and not use_dist and not args.no_extras and do_fp32 and not xnor_model:
            try:
                extras["torchrun_world1"] = torchrun_world1(args)
            except Exception as ex:
                extras["torchrun_world1"] = {"error": repr(ex)}
            # the dominant FP32 kernel's HBM traffic, measured now (the committed passes remain the fallback)
            rl = result["fp32"].get("roofline") if "fp32" in result else None
            if rl and rl.get("kernel"):
                try:
                    t_live, src = live_pmc_traffic(args, rl["kernel"], "fp32")
                except Exception as ex:
                    t_live, src = None, repr(ex)
                if t_live is not None:
                    rl["traffic_committed_passes"] = rl.get("traffic")
                    rl["traffic"], rl["traffic_source"] = t_live, src
                else:
                    rl["traffic_source"] = (rl.get("traffic_source") or "") + "; live PMC pass unavailable: %s" % src
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(cfg, wts, args.size, args.size, 0 if do_fp32 else 1, args.cpu_seconds)
                if cpu and do_fp32 and do_int8:
                    q = cpu_baseline(cfg_q, wts, args.size, args.size, 1, args.cpu_seconds / 2)
                    if q:
                        cpu["int8"] = {k: q[k] for k in ("value", "sample")}
                        if not args.no_extras and not xnor_model:
                            try:
                                cpu["int8"]["hip_int8_vs_reference_int8"] = int8_vs_reference_int8(
                                    Network, cfg_q, wts, args.size, local_rank, args.thresh, args.nms if args.nms > 0 else 0.4)
                            except Exception as ex:
                                cpu["int8"]["hip_int8_vs_reference_int8"] = {"error": repr(ex)}
            except Exception as e:      # the baseline is reported, never required
                cpu = {"error": repr(e)}
        head = result["fp32"] if do_fp32 else (result["int8"] if do_int8 else result["bf16"])
        dtype = "f32" if do_fp32 else ("i8" if do_int8 else "bf16")
        modes = ("FP32" if do_fp32 else "") + (" & INT8" if do_fp32 and do_int8 else ("INT8" if do_int8 else ""))
        if not do_fp32 and not do_int8:
            modes = "BF16 (opt-in)"
        if xnor_model:
            dtype, modes = "u1", "BIT1-XNOR"      # 1-bit operands (64-bit packed words), FP32 first/last layer
        out = {
            "metric": "images/sec (whole node) %s %dx%d batch %d %s" % (args.model, args.size, args.size,
                                                                       args.global_batch, modes),
            "value": head["value"],
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "launch": {"ranks": world, "rccl_ranks": (dist.get_world_size() if use_dist else 0),
                       "form": (os.environ.get("YL_BENCH_LAUNCH") or ("torch.distributed.run" if use_dist else "single process")),
                       "gpus_flag": args.gpus},
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%s.cfg %dx%d global batch=%d (%d/GPU) %s, synthetic weights+images resident in "
                                   "HBM, forward + on-device detection decode/compaction + NMS%s; `value` = the %s leg" % (
                                       args.model, args.size, args.size, args.global_batch, b_local, modes,
                                       " + RCCL all-gather of detections" if use_dist else "",
                                       ("BIT1-XNOR" if xnor_model else "FP32") if do_fp32 else ("INT8" if do_int8 else "BF16"))
                                   + ("; INT8 input_calibration recomputed for the synthetic weights" if (do_int8 and cfg_q != cfg) else
                                      ("; INT8 on the cfg's shipped input_calibration" if do_int8 else "")),
                       "global_batch": args.global_batch,
                       "parallelism": "image-batch sharding x%d (%s scaling)" % (world, args.scaling),
                       "gflop_per_image": head.get("gflop_per_image"),
                       "arithmetic": head.get("arithmetic")},
            "roofline": head.get("roofline"),
            "detect_ms_per_step": head.get("detect_ms_per_step"),
            "detections_per_image": head.get("detections_per_image"),
            "cpu_baseline": cpu,
            "pcie_inclusive": head.get("pcie_inclusive"),
        }
        if head.get("decode_inclusive") is not None:
            out["decode_inclusive"] = head["decode_inclusive"]
        if do_fp32 and do_int8:
            i8 = result["int8"]
            out["int8"] = {k: i8.get(k) for k in ("value", "ms_per_step", "roofline", "agreement_vs_fp32",
                                                  "detections_per_image", "detect_ms_per_step")}
            out["int8"]["unit"] = "images/sec"
            out["int8"]["speedup_vs_fp32"] = i8["value"] / head["value"]
        if do_bf16 and (do_fp32 or do_int8):
            b16 = result["bf16"]
            out["bf16"] = {k: b16.get(k) for k in ("value", "ms_per_step", "roofline", "agreement_vs_fp32",
                                                   "detections_per_image", "detect_ms_per_step")}
            out["bf16"]["unit"] = "images/sec"
            out["bf16"]["what"] = ("opt-in: yl_network_set_precision(BF16) -- bf16 operands on v_mfma_f32_32x32x16_bf16, "
                                   "FP32 accumulate; outside the 1e-4 contract, never `value`")
        out.update(extras)
        emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
