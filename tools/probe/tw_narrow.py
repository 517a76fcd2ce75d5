"""A/B of the narrow (48-channel, deep look-ahead) temporal sliding-window weight gradient on the stem's temporal
convolution (45 -> 64, 64 clips of 16 x 56 x 56), with and without the fused BatchNorm apply:
    python tools/probe/tw_narrow.py [Ci]        # Ci = 144: the layer-1 temporal convolutions (wide form either way)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H
from peppa_amd import layers as L

dev = "cuda"
CI = int(sys.argv[1]) if len(sys.argv) > 1 else 45
geom = L.ConvGeom(64, (16, 56, 56), CI, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
g = torch.Generator().manual_seed(0)
y = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(dev)
y[:, CI:] = 0
dy = torch.randn(geom.M, geom.out_cstride, generator=g).to(torch.bfloat16).to(dev)
scale = (0.5 + torch.rand(geom.in_cstride, generator=g)).to(dev)
shift = (0.3 * torch.randn(geom.in_cstride, generator=g)).to(dev)
scale[CI:] = 0
shift[CI:] = 0


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


outs = {}
for narrow, prod in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (1, 0), (0, 1), (1, 1)):
    H.set_option("tw_narrow", narrow)
    H.set_option("tw_producers", prod)
    for bn in (False, True):
        xb = (scale, shift, True) if bn else None
        t = timeit(lambda: L.conv_wgrad_raw(y, dy, geom, x_bn=xb))
        outs[(narrow | prod, bn)] = L.conv_wgrad_raw(y, dy, geom, x_bn=xb).clone()
        print(f"tw_narrow={narrow} tw_producers={prod} fused_bn={bn}: {t:7.1f} us  ({(geom.Min * geom.in_cstride * 2 + geom.M * 128) / t / 1e6:.2f} TB/s of compulsory bytes)", flush=True)
for bn in (False, True):
    a, b = outs[(0, bn)], outs[(1, bn)]
    print(f"fused_bn={bn}: max |narrow - wide| = {(a - b).abs().max().item():.3e} (scale {a.abs().max().item():.3e})")
