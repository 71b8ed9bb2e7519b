import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_bf16 as T
from oracle import dcn as odcn
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
for shape, variant, std in (((1, 32, 16, 16, 128), 4, 0.0), ((1, 32, 16, 16, 128), 4, 0.4), ((1, 128, 16, 32, 128), 4, 0.4), ((1, 32, 8, 16, 128), 3, 0.4)):
    x, wt, b, off, m, om = T._dcn_case(shape, std, 1, None)
    ref = odcn.dcn_v2_forward(x, off, m, wt, b, 1, 1, 1, 1)
    got = T._run_conv(x, wt, b, None, 1, 1, 0, None, 0, -1, 1, om, variant=variant, patch=True)
    e = (got - ref).abs()
    print(shape, variant, std, "max err", e.max().item(), "scale", ref.abs().max().item())
    print("err by row y:", e.amax(dim=(0, 1, 3)))
    print("err by col x:", e.amax(dim=(0, 1, 2)))
    print("err by channel (first 32):", e.amax(dim=(0, 2, 3))[:32])
