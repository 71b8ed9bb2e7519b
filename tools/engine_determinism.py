#!/usr/bin/env python
"""Run-to-run identity of the whole forward at the bench sizes: N forwards on the same frames, every output and every named
intermediate buffer of the plan compared bit for bit with the first forward.  A difference is reported with the first buffer (in
plan order) that shows it.  The long form of the engine-level determinism checks in tests/ (which run 2 forwards).
usage: python tools/engine_determinism.py [fp32|bf16] [N]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3dssd_amd import synth                                   # noqa: E402


def main(dtype, n):
    from model.M3d_inference_align import build
    B, crop = (64, (384, 1280)) if dtype == "bf16" else (8, (384, 1280))
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0), strict=True)
    net = net.to("cuda:0").set_compute_dtype(dtype)
    x = synth.synth_frames(B, crop, 7).to("cuda:0")
    plan = None
    ref = None
    bad = 0
    for it in range(n):
        with torch.no_grad():
            outs = net(x)[:4]
        torch.cuda.synchronize()
        if plan is None:
            plan = net.engine().plan_for(B, *crop)
        snap = {"out%d" % i: t.clone() for i, t in enumerate(outs)}
        for k, v in plan.named.items():
            t = getattr(v, "t", v)
            if torch.is_tensor(t):
                snap[k] = t.clone()
        if ref is None:
            ref = snap
            continue
        diff = [k for k in snap if not torch.equal(ref[k].view(torch.uint8), snap[k].view(torch.uint8))]
        if diff:
            bad += 1
            print("forward %d differs from forward 0 in: %s" % (it, ", ".join(diff[:12])))
    print("%s bs=%d: %d of %d forwards differ from the first (%d buffers compared)" % (dtype, B, bad, n - 1, len(ref)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fp32", int(sys.argv[2]) if len(sys.argv) > 2 else 50)
