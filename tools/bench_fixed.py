import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
B = 64
for (k, p) in (((1, 1, 1), (0, 0, 0)), ((1, 1, 3), (0, 0, 1)), ((1, 3, 3), (0, 1, 1)), ((3, 3, 3), (1, 1, 1))):
    for stats in (False, True):
        geom = L.ConvGeom(B, (16, 56, 56), 64, 144, k, (1, 1, 1), p)
        x = torch.randn(geom.Min, 64, device=dev).to(torch.bfloat16)
        w = torch.randn(144, 64, *k, device=dev) * 0.05
        wf, _ = L.prep_conv_weights(w, geom, need_dgrad=False)
        y = torch.empty(geom.M, 144, dtype=torch.bfloat16, device=dev)
        part = torch.empty(geom.nblk, 2, 144, device=dev)
        t = timeit(lambda: H.igemm(x, wf, y, geom.M, 144, geom.Kf, geom.g_fwd(), geom.Kf, 144, b_rows=144,
                                   colstats=part if stats else None, ldstat=144))
        print(f"taps {geom.taps:2d} (K={geom.Kf:4d}, {geom.Kf//64:2d} steps) stats={int(stats)}: {t:8.1f} us")
