"""Grouped transformer weight gradients (12 layers per launch, M = 7296): round 3's tile order, the flat slab-sharing order,
and the ring kernel's 128 x 256 tiles.     python tools/probe/wgrad_group_ab.py"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L
dev = "cuda"


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, NB = int(os.environ.get("M", "7296")), 12
for (N, K) in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
    items = [(torch.randn(M, K, device=dev).to(torch.bfloat16), torch.randn(M, N, device=dev).to(torch.bfloat16)) for _ in range(NB)]
    fl = 2.0 * M * N * K * NB
    line = f"N={N:5d} K={K:5d}:"
    for flat, ring in ((0, 0), (1, 0), (1, 1)):
        H.set_option("wgrad_flat", flat)
        H.set_option("wgrad_group_ring", ring)
        t = timeit(lambda: L.linear_wgrad_group(items, M, N, K))
        line += f"  flat={flat} ring={ring}: {t:7.1f} us {fl / t / 1e6:5.0f} TF |"
    print(line, flush=True)
