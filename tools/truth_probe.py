#!/usr/bin/env python
"""Where does the fp32 parity margin go?  HIP path, fp32 CPU oracle and a FLOAT64 run of the same oracle graph ("truth")
on the benched configuration (bs = 8, 1280x384), stage by stage.  All three take the engine's discrete decisions.

    python tools/truth_probe.py [B]            (GPU box; diagnostic only, not on the product path)

For every tap: max |hip - truth|, max |oracle32 - truth|, max |hip - oracle32|.  If the oracle's own distance to the
float64 result is of the order of the parity bound, the bound measures the conditioning of the synthetic network, not the
kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.set_num_threads(16)
from m3dssd_amd import synth  # noqa: E402
from model.M3d_inference_align import build  # noqa: E402
from oracle import dcn as odcn  # noqa: E402
from oracle import model_cpu  # noqa: E402


def dcn_any_dtype(inp, offset, mask, weight, bias, stride=1, pad=0, dil=1, deformable_groups=1):
    """dcn_v2_im2col_cuda.cu:18-47,129-178 + the GEMM, vectorised in torch at the dtype of `inp` (float64 for the truth run)."""
    N, C, H, W = inp.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = odcn.out_size(H, W, kh, kw, stride, pad, dil)
    dt = inp.dtype
    ys = (torch.arange(Ho, dtype=dt) * stride - pad).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=dt) * stride - pad).view(1, 1, Wo)
    out = bias.view(1, Co, 1, 1).expand(N, Co, Ho, Wo).clone()
    xf = inp.reshape(N, C, H * W)

    def corner(hh, ww, valid):
        ok = valid & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
        idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).long().view(N, 1, -1).expand(N, C, -1)
        return torch.gather(xf, 2, idx).view(N, C, Ho, Wo) * ok.unsqueeze(1).to(dt)

    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            h_im = ys + i * dil + offset[:, 2 * k]
            w_im = xs + j * dil + offset[:, 2 * k + 1]
            valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            hl, wl = torch.floor(h_im), torch.floor(w_im)
            lh, lw = (h_im - hl).unsqueeze(1), (w_im - wl).unsqueeze(1)
            val = ((1 - lh) * (1 - lw) * corner(hl, wl, valid) + (1 - lh) * lw * corner(hl, wl + 1, valid)
                   + lh * (1 - lw) * corner(hl + 1, wl, valid) + lh * lw * corner(hl + 1, wl + 1, valid))
            val = val * mask[:, k:k + 1]
            out += torch.einsum("oc,nchw->nohw", weight[:, :, i, j], val)
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    crop = (384, 1280)
    dev = torch.device("cuda:0")
    conf = synth.synth_conf(crop, 0, batch_size=B, device="cuda:0")
    sd = synth.synth_state_dict(0)
    x = synth.synth_frames(B, crop, 1234)
    x[B // 2:, :, :, (2 * crop[1]) // 3:] = 0.0
    net = build(conf, "test")
    net.load_state_dict(sd)
    net = net.to(dev)
    with torch.no_grad():
        out = [t.cpu() for t in net(x.to(dev))]
    plan = net.engine().plan_for(B, *crop)
    fh, fw = crop[0] // 8, crop[1] // 8
    ind = plan.named["sel_idx"].view(B, 1, fh, fw).long().cpu()
    hard = (plan.named["sel_prob"].view(B, 1, fh, fw).cpu() > 0.5).float()
    cconf = synth.synth_conf(crop, 0, batch_size=B, device="cpu")
    t32, t64 = {}, {}
    with torch.no_grad():
        o32 = model_cpu.rpn_forward(sd, cconf, x, t32, inject={"sel": {"ind": ind, "hard": hard}})
        # float64 run of the same graph: parameters / frames widened, DCN at float64, same injected decisions
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        real_dcn = model_cpu.D.dcn_v2_forward
        real_tab = model_cpu.shape_align_offsets
        model_cpu.D.dcn_v2_forward = dcn_any_dtype
        model_cpu.shape_align_offsets = lambda *a, **k: real_tab(*a, **k).double()
        torch.set_default_dtype(torch.float64)
        try:
            o64 = model_cpu.rpn_forward(sd64, cconf, x.double(), t64, inject={"sel": {"ind": ind, "hard": hard.double()}})
        finally:
            torch.set_default_dtype(torch.float32)
            model_cpu.D.dcn_v2_forward = real_dcn
            model_cpu.shape_align_offsets = real_tab
    print("%-16s %12s %12s %12s   (max abs; scale = max |truth|)" % ("tap", "hip-truth", "orc32-truth", "hip-orc32"))
    for name in ("level2", "level3", "level4", "level5", "feats0", "feats", "feats_align2d", "feats_align3d", "feats_gl"):
        got = plan.named[name].torch_nchw().cpu().double()
        a, b = t32[name].double(), t64[name].double()
        print("%-16s %12.3e %12.3e %12.3e   scale %.2e" % (name, (got - b).abs().max(), (a - b).abs().max(),
                                                           (got - a).abs().max(), b.abs().max()))
    for i, name in enumerate(("cls", "prob", "bbox_2d", "bbox_3d")):
        g, a, b = out[i].double(), o32[i].double(), o64[i].double()
        print("%-16s %12.3e %12.3e %12.3e   scale %.2e" % (name, (g - b).abs().max(), (a - b).abs().max(), (g - a).abs().max(),
                                                           b.abs().max()))
    names3 = ("x3d", "y3d", "z3d", "w3d", "h3d", "l3d", "rY3d")
    g, a, b = out[3].double(), o32[3].double(), o64[3].double()
    for c, n in enumerate(names3):
        print("  bbox_3d.%-6s %12.3e %12.3e %12.3e   rms hip-truth %.2e  rms orc32-truth %.2e" % (
            n, (g - b)[..., c].abs().max(), (a - b)[..., c].abs().max(), (g - a)[..., c].abs().max(),
            (g - b)[..., c].pow(2).mean().sqrt(), (a - b)[..., c].pow(2).mean().sqrt()))
    # ANAB internals at float64: how peaky is the 337-key softmax, how large are the logits?
    kp, vp = t64["anab.key_pooled"], t64["anab.value_pooled"]
    f3d = t64["feats_align3d"]
    q = torch.nn.functional.conv2d(f3d, sd64["bbox_z3d_gl.0.query_conv.weight"]).flatten(2).permute(0, 2, 1)
    logits = torch.bmm(q, kp)
    att = torch.softmax(logits, -1)
    print("ANAB logits: max |l| %.1f, mean max-prob %.3f, frac rows with max-prob > 0.9: %.3f" % (
        logits.abs().max(), att.max(-1)[0].mean(), (att.max(-1)[0] > 0.9).double().mean()))


if __name__ == "__main__":
    main()
