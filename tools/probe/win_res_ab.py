"""A/B of the window kernel's producer form on the layer-1 / layer-2 data gradients WITH the residual add (the form the
step runs for the second convolution of a block):    python tools/probe/win_res_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H
from peppa_amd import layers as L

dev = "cuda"


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


for Ci, Co, thw in ((64, 144, (16, 56, 56)), (128, 288, (8, 28, 28)), (256, 576, (4, 14, 14))):
    geom = L.ConvGeom(64, thw, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    dy = torch.randn(geom.M, geom.out_cstride, device=dev).to(torch.bfloat16)
    res = torch.randn(geom.Min, geom.in_cstride, device=dev).to(torch.bfloat16)
    wf, wd = L.prep_conv_weights(torch.randn(Co, Ci, 1, 3, 3, device=dev) * 0.05, geom)
    for rep in range(2):
        for wp in (0, 1):
            H.set_option("win_producers", wp)
            t0 = timeit(lambda: L.conv_dgrad(dy, geom, wd))
            t1 = timeit(lambda: L.conv_dgrad(dy, geom, wd, residual=res))
            print(f"{Co}->{Ci} win_producers={wp}: plain {t0:7.1f} us, with residual {t1:7.1f} us", flush=True)
