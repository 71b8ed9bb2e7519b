#!/usr/bin/env python
"""Per-tile sampling radius of the DCNv2 layers of the synthetic network (bf16 engine, bs 8, 1280x384): for every 3x3 DCN layer,
the histogram of ceil(max |offset|) over 8x16 pixel patches (one wave of a wave-tile kernel) and over 16x32 tiles (a 4-wave
workgroup) -- what decides how large an LDS sampling window has to be (csrc/bf16_dcn_patch.hip).
    python tools/dcn_radius_hist.py [B]"""
import sys

import torch

sys.path.insert(0, ".")
from m3dssd_amd import synth                       # noqa: E402
from model.M3d_inference_align import build        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
crop = (384, 1280)
conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
net = build(conf, "test")
net.load_state_dict(synth.synth_state_dict(0), strict=True)
net = net.to("cuda:0").set_compute_dtype("bf16")
x = synth.synth_frames(B, crop, 1234).to("cuda:0")
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
plan = net.engine().plan_for(B, *crop)
for name, v in plan.named.items():
    if not name.endswith(".om"):
        continue
    om = v.t.view(v.n, v.h, v.w, v.cs)[..., :18].abs().amax(-1)          # [B, H, W] max |offset| per pixel
    line = "%-34s %3dx%-3d  max %.2f  mean |off|max/px %.2f" % (name, v.h, v.w, float(om.max()), float(om.mean()))
    for th, tw in ((8, 16), (16, 32), (8, 32), (24, 16)):
        if v.h % th or v.w % tw:
            continue
        t = om.view(v.n, v.h // th, th, v.w // tw, tw).amax((2, 4)).ceil().long().flatten()
        hist = torch.bincount(t, minlength=12)[:12].tolist()
        cum = torch.cumsum(torch.tensor(hist, dtype=torch.float64), 0) / max(1, t.numel())
        line += "\n      %2dx%-2d tiles: radius hist %s  cum %s" % (th, tw, hist, ["%.2f" % c for c in cum.tolist()])
    print(line)
