import ctypes, sys, torch
sys.path.insert(0, ".")
from m3dssd_amd import _hip
from m3dssd_amd.engine_bf16 import pack_conv_bf16
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SEED_MODE = sys.argv[2] if len(sys.argv) > 2 else "fixed"        # "test" = the seeds of tests/test_gpu_bf16.py
dev = torch.device("cuda:0")
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
def same(a, b):
    """bit patterns, so that a NaN equals itself"""
    return torch.equal(a.view(torch.int16), b.view(torch.int16))


def analyze(cin, cout, H, W, B, k, x, wf, om, good, badout, px_list):
    """For pixels whose output differs between two launches: express (bad - good), before the activation, in the per-tap
    contributions C_t = W_t . sample_t of that pixel.  A dropped tap shows as coefficient -1 on one C_t."""
    import math
    kk, pad = k * k, k // 2
    xf = x.float().cpu().view(B, H, W, -1)[..., :cin]
    omc = om.cpu()
    for px in px_list:
        n, rem = divmod(int(px), H * W)
        h, w = divmod(rem, W)
        def tap_sample(t, dh, dw, mk):
            ti, tj = divmod(t, k)
            hi, wi = h - pad + ti + dh, w - pad + tj + dw
            s = torch.zeros(cin)
            if hi > -1 and wi > -1 and hi < H and wi < W:
                hl, wl = math.floor(hi), math.floor(wi)
                lh, lw = hi - hl, wi - wl
                for (hh, ww, wt) in [(hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw), (hl + 1, wl, lh * (1 - lw)),
                                     (hl + 1, wl + 1, lh * lw)]:
                    if 0 <= hh < H and 0 <= ww < W:
                        s += wt * mk * xf[n, hh, ww]
            return s.to(torch.bfloat16).float(), wf[:, :, ti, tj].to(torch.bfloat16).float()

        C, alt = [], []
        for t in range(kk):
            dh, dw, mk = float(omc[px, 2 * t]), float(omc[px, 2 * t + 1]), float(omc[px, 2 * kk + t])
            s, wt_ = tap_sample(t, dh, dw, mk)
            for c0 in range(0, cin, 64):                                # one regressor per K-step (64 channels of one tap)
                C.append(wt_[:, c0:c0 + 64] @ s[c0:c0 + 64])
            # hypothesis "stale offsets": tap t sampled with the (dh, dw) of tap t - 1 (zeros for tap 0), own mask
            pdh, pdw = (float(omc[px, 2 * t - 2]), float(omc[px, 2 * t - 1])) if t else (0.0, 0.0)
            s2, _ = tap_sample(t, pdh, pdw, mk)
            pmk = float(omc[px, 2 * kk + t - 1]) if t else 0.0
            s3, _ = tap_sample(t, dh, dw, pmk)                          # hypothesis "stale mask"
            s4, _ = tap_sample(t, pdh, pdw, pmk)                        # both stale
            alt.append((wt_ @ (s2 - s), wt_ @ (s3 - s), wt_ @ (s4 - s)))
        C = torch.stack(C, 1)                                           # [cout, kk * cin / 64]
        inv = lambda y: torch.where(y >= 0, y, y / 0.01)                # LeakyReLU(0.01) inverse
        g, b_ = inv(good[px].float().cpu()), inv(badout[px].float().cpu())
        ref = C.sum(1)
        sol = torch.linalg.lstsq(C, (b_ - g).unsqueeze(1)).solution.squeeze(1)
        best = min(((float((b_ - ref - a_[j]).abs().max()), t, ("offsets", "mask", "both")[j]) for t, a_ in enumerate(alt) for j in range(3)))
        print("   best stale-input hypothesis: tap %d stale %s -> residual %.3f" % (best[1], best[2], best[0]))
        nk = cin // 64
        tw = max(range(kk), key=lambda t: sum(abs(float(sol[t * nk + j])) for j in range(nk)))
        ti, tj = divmod(tw, k)
        dh, dw, mk = float(omc[px, 2 * tw]), float(omc[px, 2 * tw + 1]), float(omc[px, 2 * kk + tw])
        hi, wi = h - pad + ti + dh, w - pad + tj + dw
        hl, wl = math.floor(hi), math.floor(wi)
        lh, lw = hi - hl, wi - wl
        wt_ = wf[:, :, ti, tj].to(torch.bfloat16).float()
        cq = []
        for (hh, ww, wq) in [(hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw), (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)]:
            cq.append(wt_ @ (wq * mk * xf[n, hh, ww]) if 0 <= hh < H and 0 <= ww < W else torch.zeros(cout))
        cq = torch.stack(cq, 1)
        sq = torch.linalg.lstsq(cq, (b_ - g).unsqueeze(1)).solution.squeeze(1)
        res = float(((b_ - g) - cq @ sq).abs().max())
        print("   tap %d (h_im %.2f w_im %.2f mask %.2f): (bad - good) in its 4 corner contributions: %s  residual %.3f" % (
            tw, hi, wi, mk, " ".join("%+.2f" % v for v in sq.tolist()), res))
        print("   pixel %d (img %d, %d, %d): |good - ref| %.3f  |bad - ref| %.3f  (bad - good) in K-step contributions (tap-major): %s" % (
            px, n, h, w, float((g - ref).abs().max()), float((b_ - ref).abs().max()), " ".join("%+.2f" % v for v in sol.tolist())))


for (cin, cout, H, W, B, k, cs) in [(128, 128, 48, 160, 64, 3, 128), (256, 256, 24, 80, 64, 3, 256), (128, 128, 48, 160, 64, 1, 128), (64, 64, 33, 47, 5, 3, 72)]:
    g = torch.Generator().manual_seed(cin + k if SEED_MODE == "test" else 1)
    x = torch.randn(B * H * W, cs, generator=g).to(torch.bfloat16).to(dev)
    wf = torch.randn(cout, cin, k, k, generator=g) / (k * k * cin) ** 0.5
    wp, kpad = pack_conv_bf16(wf, None, None, dev)
    kk = k * k
    om = torch.cat([torch.randn(B * H * W, 2 * kk, generator=g) * 2.0, torch.rand(B * H * W, kk, generator=g), torch.zeros(B * H * W, 32 - 3 * kk)], 1).contiguous().to(dev)
    outs = []
    for rep in range(REPS):
        out = torch.zeros(B * H * W, cout, device=dev, dtype=torch.bfloat16)
        d = _hip.ConvBf16Desc()
        d.inp, d.in_cs, d.N, d.H, d.W, d.Cin = x.data_ptr(), cs, B, H, W, cin
        d.wgt, d.Cout, d.Cout_pad, d.Kpad = wp.data_ptr(), cout, wp.shape[0], kpad
        d.kh = d.kw = k
        d.stride, d.pad, d.Ho, d.Wo = 1, k // 2, H, W
        d.out, d.out_cs, d.out_mode, d.act, d.sigmoid_from, d.groups = out.data_ptr(), cout, 0, 1, -1, 1
        d.dcn_offmask, d.dcn_om_cs = om.data_ptr(), 32
        assert L.m3d_conv_bf16_forward(ctypes.byref(d), st) == 0
        torch.cuda.synchronize()
        outs.append(out)
    # reference launch = the first one that some other launch reproduces bit for bit
    gi = 0
    for a_ in range(min(REPS, 6)):
        if any(same(outs[a_], outs[b_]) for b_ in range(REPS) if b_ != a_):
            gi = a_
            break
    good = outs[gi]
    bad = [i for i in range(REPS) if i != gi and not same(good, outs[i])]
    print(cin, cout, H, W, B, k, "launches differing from the reference launch: %d of %d" % (len(bad), REPS - 1))
    for i in bad[:6]:
        dd = (good.float() - outs[i].float()).abs().nan_to_num(0.0)
        nz = (good.view(torch.int16) != outs[i].view(torch.int16)).nonzero()
        px = nz[:, 0].unique()
        print("   launch %d: %d values in %d pixels, max |diff| %.4f; tile %s row-in-tile %s channels %d..%d" % (
            i, nz.shape[0], px.numel(), float(dd.max()), (px // 128).unique()[:6].tolist(), (px % 128)[:16].tolist(),
            int(nz[:, 1].min()), int(nz[:, 1].max())))
        if "analyze" in sys.argv:
            analyze(cin, cout, H, W, B, k, x, wf, om, good, outs[i], px[:4].tolist())
