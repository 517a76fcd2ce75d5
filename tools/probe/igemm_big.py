"""256 x 256 forward / data-gradient tiles (pp_set_option igemm_big): check against the ring / gather kernels, then time.
    python tools/probe/igemm_big.py"""
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


def both(fn):
    H.set_option("igemm_big", 0); ref = fn()
    H.set_option("igemm_big", -1); got = fn()
    H.set_option("igemm_big", 0)
    return ref, got


torch.manual_seed(0)
# ---- conv forward with GELU + saved pre-activation, strided data gradient (row map), k = 2 and k = 3, ragged M
for name, B, k, st, T in (("k3s2", 5, 3, 2, 9999), ("k2s2", 7, 2, 2, 7001), ("k3s2 odd", 3, 3, 2, 16667)):
    geom = L.ConvGeom(B, (T, 1, 1), 512, 512, (k, 1, 1), (st, 1, 1), (0, 0, 0))
    x = torch.randn(geom.Min, 512, device="cuda").bfloat16()
    dy = torch.randn(geom.M, 512, device="cuda").bfloat16()
    w = torch.randn(512, 512, k, 1, 1, device="cuda") * 0.03
    wf, wd = L.prep_conv_weights(w, geom)

    def fwd():
        pre = L.empty((geom.M, 512), torch.bfloat16, x)
        y, _ = L.conv_fwd(x, geom, wf, act=H.ACT_GELU, pre=pre)
        return y.clone(), pre.clone()
    (y0, p0), (y1, p1) = both(fwd)
    d0, d1 = both(lambda: L.conv_dgrad(dy, geom, wd).clone())
    print(f"audio conv {name} M={geom.M}: fwd max|d| {(y1.float() - y0.float()).abs().max().item():.3e} pre {(p1.float() - p0.float()).abs().max().item():.3e} "
          f"bitwise {torch.equal(y0, y1) and torch.equal(p0, p1)}; dgrad max|d| {(d1.float() - d0.float()).abs().max().item():.3e} bitwise {torch.equal(d0, d1)}", flush=True)
# ---- conv forward with padding (3-D, strided), plain epilogue
geom = L.ConvGeom(8, (8, 56, 56), 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1))
x = torch.randn(geom.Min, geom.in_cstride, device="cuda").bfloat16()
w = torch.randn(230, 64, 1, 3, 3, device="cuda") * 0.05
wf, wd = L.prep_conv_weights(w, geom)
y0, y1 = both(lambda: L.conv_fwd(x, geom, wf)[0].clone())
print(f"l2.0 spatial s2 fwd (no statistics) M={geom.M}: max|d| {(y1.float() - y0.float()).abs().max().item():.3e} bitwise {torch.equal(y0, y1)}", flush=True)
# ---- dense with the fused epilogue
M, N, K = 50000, 768, 520
xd = torch.randn(M, K + 8, device="cuda").bfloat16()[:, :K].contiguous()
wl = torch.randn(N, K, device="cuda") * 0.05
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").bfloat16()
wfl, wtl = L.prep_linear(wl)
xp = torch.zeros(M, wfl.shape[1], device="cuda", dtype=torch.bfloat16); xp[:, :K] = xd
for kw in (dict(), dict(bias=bias, act=H.ACT_GELU), dict(bias=bias, residual=res, dropout=(0.1, 1234))):
    def lin():
        pre = L.empty((M, N), torch.bfloat16, xp) if "act" in kw else None
        y = L.linear_fwd(xp, M, wfl, N, pre=pre, **kw)
        return (y.clone(), pre.clone() if pre is not None else None)
    (a0, q0), (a1, q1) = both(lin)
    print(f"dense M={M} N={N} K={K} {sorted(kw)}: max|d| {(a1.float() - a0.float()).abs().max().item():.3e} bitwise {torch.equal(a0, a1) and (q0 is None or torch.equal(q0, q1))}", flush=True)
# ---- timing at the step's shapes
B = 64
for name, k, st, T in (("conv1 k3s2 T=7359", 3, 2, 7359), ("conv2 k3s2 T=3679", 3, 2, 3679), ("conv3 k3s2 T=1839", 3, 2, 1839), ("conv4 k3s2 T=919", 3, 2, 919),
                       ("conv5 k2s2 T=459", 2, 2, 459), ("conv6 k2s2 T=229", 2, 2, 229)):
    geom = L.ConvGeom(B, (T, 1, 1), 512, 512, (k, 1, 1), (st, 1, 1), (0, 0, 0))
    x = torch.randn(geom.Min, 512, device="cuda").bfloat16()
    dy = torch.randn(geom.M, 512, device="cuda").bfloat16()
    w = torch.randn(512, 512, k, 1, 1, device="cuda") * 0.03
    wf, wd = L.prep_conv_weights(w, geom)
    pre = L.empty((geom.M, 512), torch.bfloat16, x)
    fl = 2.0 * geom.M * 512 * k * 512
    t = {}
    for big in (0, 1):
        H.set_option("igemm_big", -big)
        t[big] = (timeit(lambda: L.conv_fwd(x, geom, wf, act=H.ACT_GELU, pre=pre)), timeit(lambda: L.conv_dgrad(dy, geom, wd)))
    H.set_option("igemm_big", 0)
    print(f"{name:20s} M={geom.M:7d}: fwd {t[0][0]:7.1f} -> {t[1][0]:7.1f} us ({fl/t[1][0]/1e6:5.0f} TF) | dgrad {t[0][1]:7.1f} -> {t[1][1]:7.1f} us ({fl/t[1][1]/1e6:5.0f} TF)", flush=True)
