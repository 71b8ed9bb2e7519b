"""Times m3d_refine_3d on a batch of 8 x 40 detections against the CPU restatement of the reference's per-box loop."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd.host import refine as HR          # noqa: E402
from oracle import refine as R                    # noqa: E402

g = np.load("tests/golden/refine.npz")
p2, rows = g["p2"], g["rows"]
dets = np.stack([rows[:40]] * 8).astype(np.float32)
dets[:, :, 4] = 0.9
dev = torch.device("cuda:0")
d = torch.from_numpy(dets).to(dev)
c = torch.full((8,), 40, dtype=torch.int32, device=dev)
for _ in range(3):
    HR.refine_detections(d, c, p2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = HR.refine_detections(d, c, p2)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / 20 * 1e3
p2i = np.linalg.inv(p2)
t0 = time.perf_counter()
for b in range(8):
    for r in dets[b]:
        R.refine_row(r, p2, p2i)
cpu_ms = (time.perf_counter() - t0) * 1e3
print("refine 8 x 40 detections: device %.3f ms per batch (incl. host inverse + upload), CPU restatement %.1f ms" % (gpu_ms, cpu_ms))
