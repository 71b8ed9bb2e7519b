#!/usr/bin/env python
"""Device NMS alone: B images x n boxes through m3d_nms_sorted_dev, HIP-event time per call (M3D_NMS_DIV=1: the division form).
    python tools/nms_bench.py [B] [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from m3dssd_amd import synth  # noqa: E402
from m3dssd_amd.host import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
dev = torch.device("cuda:0")
dets = np.stack([synth.synth_boxes(n, seed=100 + i) for i in range(B)])
dets = np.stack([d[np.argsort(-d[:, 4], kind="stable")] for d in dets])
x = torch.from_numpy(dets).to(dev)
for _ in range(3):
    keep, num = ops.nms_sorted(x, 0.4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    keep, num = ops.nms_sorted(x, 0.4)
e1.record()
torch.cuda.synchronize()
print("B %d n %d: %.3f ms per call, kept %d, form %s" % (B, n, e0.elapsed_time(e1) / 20, int(num.sum()), os.environ.get("M3D_NMS_DIV", "0")))
