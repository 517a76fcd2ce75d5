"""The paired-pixel stem's forward as a window kernel (pp_stem_pairs_fwd) against the gather kernel, then timed.
    python tools/probe/stem_window.py"""
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
for B, T, Hh, W, Co in ((2, 3, 112, 112, 45), (1, 2, 64, 64, 45), (3, 1, 30, 50, 45), (2, 2, 112, 112, 48)):
    geom = L.ConvGeom.paired_stem(B, (T, Hh, W), 3, Co, (1, 7, 7), (1, 2, 2), (0, 3, 3))
    x = torch.randn(B * T * Hh * W, 4, device="cuda").bfloat16()
    x[:, 3] = 0
    x = x.view(-1, 8)
    w = torch.randn(Co, 3, 1, 7, 7, device="cuda") * 0.05
    wf, _ = L.prep_conv_weights(w, geom, need_dgrad=False)
    L.STEM_WINDOW = False
    y0, p0 = L.conv_fwd(x, geom, wf, stats=True)
    L.STEM_WINDOW = True
    assert L._stem_window_ok(geom), "window form not taken"
    y1, p1 = L.conv_fwd(x, geom, wf, stats=True)
    nc = (Co + 7) // 8 * 8
    s0, s1 = p0.double().sum(0), p1.double().sum(0)
    yf = y1[:, :Co].float()
    print(f"B={B} T={T} {Hh}x{W} Co={Co}: y bitwise {torch.equal(y0[:, :nc], y1[:, :nc])}; statistics rel diff {((s0 - s1).abs().max() / s0.abs().max()).item():.2e} "
          f"(rows {p0.shape[0]} -> {p1.shape[0]}); sum(y) vs stats {(yf.double().sum(0) - s1[0, :Co]).abs().max().item():.3e} of {s1[0, :Co].abs().max().item():.1f}", flush=True)
for B, T, Hh, W, Co in ((2, 3, 112, 112, 45), (1, 2, 64, 64, 45), (3, 1, 30, 50, 45), (2, 2, 112, 112, 48)):
    geom = L.ConvGeom.paired_stem(B, (T, Hh, W), 3, Co, (1, 7, 7), (1, 2, 2), (0, 3, 3))
    x = torch.randn(B * T * Hh * W // 2, 8, device="cuda").bfloat16()
    dy = torch.randn(geom.M, geom.out_cstride, device="cuda").bfloat16()
    L.STEM_WINDOW = False; g0 = L.conv_wgrad_raw(x, dy, geom).clone()
    L.STEM_WINDOW = True; g1 = L.conv_wgrad_raw(x, dy, geom).clone()
    print(f"wgrad B={B} T={T} {Hh}x{W} Co={Co}: max|window - gather| {(g1 - g0).abs().max().item():.3e} of {g0.abs().max().item():.1f}", flush=True)
B, T, Hh, W, Co = 64, 16, 112, 112, 45
geom = L.ConvGeom.paired_stem(B, (T, Hh, W), 3, Co, (1, 7, 7), (1, 2, 2), (0, 3, 3))
x = torch.randn(B * T * Hh * W // 2, 8, device="cuda").bfloat16()
w = torch.randn(Co, 3, 1, 7, 7, device="cuda") * 0.05
wf, _ = L.prep_conv_weights(w, geom, need_dgrad=False)
dyb = torch.randn(geom.M, geom.out_cstride, device="cuda").bfloat16()
for rep in range(2):
    for on in (False, True):
        L.STEM_WINDOW = on
        t = timeit(lambda: L.conv_fwd(x, geom, wf, stats=True))
        tw = timeit(lambda: L.conv_wgrad_raw(x, dyb, geom))
        print(f"stem B=64 16x112x112, window kernels {on}: forward {t:7.1f} us, weight gradient {tw:7.1f} us", flush=True)
