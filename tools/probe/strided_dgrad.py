"""Where do the parity classes of a stride-2 data gradient spend their time?  Times every class of layer 2.0's spatial
conv (64 <- 230, stride (1,2,2)) alone, with / without the residual add and the output row map."""
import itertools, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 64
geom = L.ConvGeom(B, (16, 56, 56), 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1))
dy = torch.randn(geom.M, geom.out_cstride, device="cuda").to(torch.bfloat16)
w = torch.randn(230, 64, 1, 3, 3, device="cuda") * 0.05
_, wd = L.prep_conv_weights(w, geom)
res = torch.randn(geom.Min, geom.in_cstride, device="cuda").to(torch.bfloat16)
dx = torch.empty(geom.Min, geom.in_cstride, device="cuda", dtype=torch.bfloat16)
kt, kh, kw = geom.k
print("whole strided dgrad:", f"{timeit(lambda: L.conv_dgrad(dy, geom, wd)):.1f} us;  with residual {timeit(lambda: L.conv_dgrad(dy, geom, wd, residual=res)):.1f} us")
for cls in L._parity_classes(geom):
    (qt, tt, ct, Rt), (qh, th, ch, Rh), (qw, tw, cw, Rw) = cls
    sel = [(a * kh + b) * kw + c for a in tt for b in th for c in tw]
    wsel = torch.empty(geom.Ci, len(sel), geom.cg_out, device="cuda", dtype=torch.bfloat16)
    H.select_taps(wd, wsel, geom.Ci, geom.taps, geom.cg_out, sel)
    g = H.gather_conv(H.CONV_DGRAD, (Rt, Rh, Rw), (geom.To, geom.Ho, geom.Wo), (len(tt), len(th), len(tw)), (1, 1, 1), (ct, ch, cw), geom.cg_out, geom.out_cstride)
    K = len(sel) * geom.cg_out
    M = geom.B * Rt * Rh * Rw
    omap = ((geom.Ti, geom.Hi, geom.Wi), geom.s, (qt, qh, qw))
    compact = torch.empty(M, geom.in_cstride, device="cuda", dtype=torch.bfloat16)
    t_full = timeit(lambda: H.igemm(dy, wsel, dx, M, geom.in_cstride, K, g, K, geom.in_cstride, b_rows=geom.Ci, residual=res, ldr=geom.in_cstride, omap=omap))
    t_nores = timeit(lambda: H.igemm(dy, wsel, dx, M, geom.in_cstride, K, g, K, geom.in_cstride, b_rows=geom.Ci, omap=omap))
    t_plain = timeit(lambda: H.igemm(dy, wsel, compact, M, geom.in_cstride, K, g, K, geom.in_cstride, b_rows=geom.Ci))
    fl = 2.0 * M * 64 * K
    print(f"class q=({qt},{qh},{qw}) taps {len(sel)} K={K:5d} M={M}: omap+res {t_full:7.1f} us | omap {t_nores:7.1f} us | compact plain {t_plain:7.1f} us "
          f"({fl / t_plain / 1e6:.0f} TF/s)")
