"""FP32 yolov3-608 at N images per GPU (default 8): step time and per-layer table, split K on / off.  usage: python tools/b8_layers.py [split_k 0|1] [batch]"""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import common  # noqa: E402
from common import Network  # noqa: E402

sk = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg, wts = common.model_files("yolov3", 608, 608)
net = Network.load(cfg, wts, B, 0, device=0, fuse=True, split_k=sk)
x = torch.rand((B, 3, 608, 608), device="cuda:0")
for _ in range(3):
    net.forward_device(x.data_ptr())
net.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    net.forward_device(x.data_ptr())
net.synchronize()
t = (time.perf_counter() - t0) / 30
nsplit = sum(",split" in net.layer_kernel(i) for i in range(net.n))
print("split_k=%d batch %d: %.3f ms per step = %.1f img/s, %d split layers" % (sk, B, t * 1e3, B / t, nsplit))
for k in range(3):
    net.forward_timed(x.data_ptr(), k)
net.synchronize()
ms, _ = net.layer_times(2)
for i in range(net.n):
    if net.layer_kernel(i):
        print(i, net.layer_kernel(i), "%.3f" % ms[i])
