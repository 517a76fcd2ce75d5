"""Layer-1 spatial data gradient (144 -> 64, M = 3.2 M rows) with and without the fused residual add."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

geom = L.ConvGeom(64, (16, 56, 56), 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1))
dy = torch.randn(geom.M, geom.out_cstride, device="cuda").to(torch.bfloat16)
res = torch.randn(geom.Min, geom.in_cstride, device="cuda").to(torch.bfloat16)
_, wd = L.prep_conv_weights(torch.randn(144, 64, 1, 3, 3, device="cuda") * 0.05, geom)
for _ in range(3):
    print(f"plain {timeit(lambda: L.conv_dgrad(dy, geom, wd)):.1f} us | with residual {timeit(lambda: L.conv_dgrad(dy, geom, wd, residual=res)):.1f} us")
a = L.conv_dgrad(dy, geom, wd, residual=res).float()
b = L.conv_dgrad(dy, geom, wd).float() + res.float()
print("max |fused - (plain + res)| =", (a - b).abs().max().item(), "of", b.abs().max().item())
