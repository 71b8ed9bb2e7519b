import ctypes, sys, torch
sys.path.insert(0, ".")
from m3dssd_amd import _hip
from m3dssd_amd.engine_bf16 import pack_conv_bf16
dev = torch.device("cuda:0")
L = _hip.lib()
st = torch.cuda.current_stream().cuda_stream
for (cin, cout, H, W, B, k, cs) in [(128, 128, 48, 160, 64, 3, 128), (256, 256, 24, 80, 64, 3, 256), (128, 128, 48, 160, 64, 1, 128), (64, 64, 33, 47, 5, 3, 72)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * H * W, cs, generator=g).to(torch.bfloat16).to(dev)
    wp, kpad = pack_conv_bf16(torch.randn(cout, cin, k, k, generator=g) / (k * k * cin) ** 0.5, None, None, dev)
    kk = k * k
    om = torch.cat([torch.randn(B * H * W, 2 * kk, generator=g) * 2.0, torch.rand(B * H * W, kk, generator=g), torch.zeros(B * H * W, 32 - 3 * kk)], 1).contiguous().to(dev)
    outs = []
    for rep in range(4):
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
    diffs = [(outs[0].float() - o.float()).abs() for o in outs[1:]]
    print(cin, cout, H, W, B, k, "max run-to-run diff", [float(dd.max()) for dd in diffs], "n diff", [int((dd > 0).sum()) for dd in diffs])
    for dd in diffs:
        nz = (dd > 0).nonzero()
        if nz.numel():
            px = nz[:, 0].unique()
            print("   differing pixels:", px.numel(), "mod128", (px % 128)[:12].tolist())
