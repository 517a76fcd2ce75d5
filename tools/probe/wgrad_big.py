"""256 x 256 weight-gradient tiles (pp_set_option wgrad_big): check against the 128 x 128 kernel, then time.
    python tools/probe/wgrad_big.py"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H, layers as L


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
H.set_option("sw_wgrad", 0)      # (keep the sliding-window kernels out of the comparison)
CONV = [("audio conv k3s2", 2, 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1001, 1, 1)),
        ("audio conv k2s2", 3, 512, 512, (2, 1, 1), (2, 1, 1), (0, 0, 0), (459, 1, 1)),
        ("l2.0 spatial s2 64->230", 2, 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 28, 28)),
        ("l3 temporal 576->256", 2, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 7, 9)),
        ("l4.0 spatial s2 256->921", 2, 256, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 14, 14))]
for name, B, Ci, Co, k, st, pd, thw in CONV:
    geom = L.ConvGeom(B, thw, Ci, Co, k, st, pd)
    x = torch.randn(geom.Min, geom.in_cstride, device="cuda").bfloat16()
    dy = torch.randn(geom.M, geom.out_cstride, device="cuda").bfloat16()
    H.set_option("wgrad_big", 0)
    ref = L.conv_wgrad_raw(x, dy, geom).clone()
    H.set_option("wgrad_big", 1)
    got = L.conv_wgrad_raw(x, dy, geom).clone()
    H.set_option("wgrad_big", 0)
    print(f"{name:28s} M={geom.M:7d}: max|big - 128| = {(got - ref).abs().max().item():.3e}  (|ref| max {ref.abs().max().item():.2f})", flush=True)
for (M, N, K) in ((1000, 768, 768), (7296, 2304, 768), (3000, 520, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    H.set_option("wgrad_big", 0); ref = L.linear_wgrad(x, dy, M, N, K, want_bias=False)[0].clone()
    H.set_option("wgrad_big", 1); got = L.linear_wgrad(x, dy, M, N, K, want_bias=False)[0].clone()
    items = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(M, N, device="cuda").bfloat16()) for _ in range(3)]
    gb = [t[0].clone() for t in L.linear_wgrad_group(items, M, N, K, want_bias=False)]
    H.set_option("wgrad_big", 0)
    gr = [t[0].clone() for t in L.linear_wgrad_group(items, M, N, K, want_bias=False)]
    H.set_option("wgrad_big", 1); wb = [(a.clone(), b.clone()) for a, b in L.linear_wgrad_group(items, M, N, K, want_bias=True)]
    H.set_option("wgrad_big", 0); wr = [(a.clone(), b.clone()) for a, b in L.linear_wgrad_group(items, M, N, K, want_bias=True)]
    db = max((a[1] - b[1]).abs().max().item() for a, b in zip(wb, wr)); dwb = max((a[0] - b[0]).abs().max().item() for a, b in zip(wb, wr))
    print(f"dense M={M} N={N} K={K}: max|big - 128| = {(got - ref).abs().max().item():.3e}; grouped x3: {max((a - b).abs().max().item() for a, b in zip(gb, gr)):.3e}; with bias: dW {dwb:.3e} db {db:.3e} (|db| max {wr[0][1].abs().max().item():.1f})  (|ref| max {ref.abs().max().item():.1f})", flush=True)
# ---- timing at the step's shapes
B = 64
TIMED = [("audio conv1 k3s2 T=7359", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (7359, 1, 1)),
         ("audio conv2 k3s2 T=3679", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (3679, 1, 1)),
         ("audio conv3 k3s2 T=1839", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1839, 1, 1)),
         ("audio conv5 k2s2 T=459", 512, 512, (2, 1, 1), (2, 1, 1), (0, 0, 0), (459, 1, 1)),
         ("l2.0 spatial 64->230 s2", 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (16, 56, 56)),
         ("l3.0 spatial 128->460 s2", 128, 460, (1, 3, 3), (1, 2, 2), (0, 1, 1), (8, 28, 28)),
         ("l4.0 spatial 256->921 s2", 256, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 14, 14)),
         ("l2.0 temporal 230->128 s2", 230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (16, 28, 28)),
         ("l3 temporal 576->256", 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 14, 14)),
         ("l4 temporal 1152->512", 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 7, 7))]
H.set_option("sw_wgrad", 1)
for name, Ci, Co, k, st, pd, thw in TIMED:
    geom = L.ConvGeom(B, thw, Ci, Co, k, st, pd)
    x = torch.randn(geom.Min, geom.in_cstride, device="cuda").bfloat16()
    dy = torch.randn(geom.M, geom.out_cstride, device="cuda").bfloat16()
    fl = 2.0 * geom.M * Co * geom.taps * Ci
    H.set_option("wgrad_big", 0); t0 = timeit(lambda: L.conv_wgrad_raw(x, dy, geom))
    H.set_option("wgrad_big", 1); t1 = timeit(lambda: L.conv_wgrad_raw(x, dy, geom))
    H.set_option("wgrad_big", 0)
    print(f"{name:28s} M={geom.M:8d}: 128x128 {t0:7.1f} us {fl/t0/1e6:6.0f} TF | 256x256 {t1:7.1f} us {fl/t1/1e6:6.0f} TF", flush=True)
M = 64 * 114
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    items = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(M, N, device="cuda").bfloat16()) for _ in range(12)]
    fl = 12 * 2.0 * M * N * K
    H.set_option("wgrad_big", 0); t0 = timeit(lambda: L.linear_wgrad_group(items, M, N, K, want_bias=False))
    H.set_option("wgrad_big", 1); t1 = timeit(lambda: L.linear_wgrad_group(items, M, N, K, want_bias=False))
    H.set_option("wgrad_big", 0); tb = timeit(lambda: L.linear_wgrad_group(items, M, N, K, want_bias=True))
    H.set_option("wgrad_big", 1); tb1 = timeit(lambda: L.linear_wgrad_group(items, M, N, K, want_bias=True))
    H.set_option("wgrad_big", 0)
    print(f"grouped x12 N={N} K={K}: 128x128 {t0:7.1f} us {fl/t0/1e6:6.0f} TF (with bias {tb:7.1f}) | 256x256 {t1:7.1f} us {fl/t1/1e6:6.0f} TF (with bias {tb1:7.1f})", flush=True)
