#!/usr/bin/env python
"""Print the parity margins of the whole network vs the CPU oracle (same discrete decisions injected)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(16)
from m3dssd_amd import synth
from model.M3d_inference_align import build
from oracle import model_cpu

for crop, B, pad in (((128, 320), 2, False), ((384, 1280), 1, True)):
    dev = torch.device("cuda:0")
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    sd = synth.synth_state_dict(0)
    x = synth.synth_frames(B, crop, 1234, pad_right_third=pad)
    net = build(conf, "test"); net.load_state_dict(sd); net = net.to(dev)
    with torch.no_grad():
        out = [t.cpu() for t in net(x.to(dev))]
    plan = net.engine().plan_for(B, crop[0], crop[1])
    fh, fw = crop[0] // 8, crop[1] // 8
    ind = plan.named["sel_idx"].view(B, 1, fh, fw).long().cpu()
    hard = (plan.named["sel_prob"].view(B, 1, fh, fw).cpu() > 0.5).float()
    cconf = synth.synth_conf(crop, 0, batch_size=B, device="cpu")
    taps = {}
    with torch.no_grad():
        o = model_cpu.rpn_forward(sd, cconf, x, taps, inject={"sel": {"ind": ind, "hard": hard}})
    print("crop", crop, "wino" if os.environ.get("M3D_WINO", "1") != "0" else "direct")
    for name in ("level2", "level3", "level4", "level5", "feats0", "feats", "feats_align2d", "feats_align3d", "feats_gl"):
        got = plan.named[name].torch_nchw().cpu()
        e = (got - taps[name]).abs()
        print("  %-14s max abs %.2e  rel %.2e" % (name, e.max(), (e / (1 + taps[name].abs())).max()))
    for i, name in enumerate(("cls", "prob", "bbox_2d", "bbox_3d")):
        print("  %-14s max abs %.2e" % (name, (out[i] - o[i]).abs().max()))
