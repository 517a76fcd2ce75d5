"""BatchNorm apply + ReLU of the temporal conv's input done on the LDS windows of the temporal window kernel and of the
temporal sliding-window weight gradient (pp_igemm a_bn_*, pp_wgrad x_bn_*: z never materialised) against the separate
pp_bn_apply pass + the plain kernels.  Layer-1 shapes, B = 64."""
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

geom = L.ConvGeom(64, (16, 56, 56), 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
M, Cp = geom.Min, geom.in_cstride
y = torch.randn(M, Cp, device="cuda").to(torch.bfloat16)
scale = torch.randn(Cp, device="cuda") * 0.5 + 1.0
shift = torch.randn(Cp, device="cuda") * 0.3
wf, _ = L.prep_conv_weights(torch.randn(64, 144, 3, 1, 1, device="cuda") * 0.05, geom)
z = torch.empty_like(y)
out = torch.empty(geom.M, geom.out_cstride, device="cuda", dtype=torch.bfloat16)

def separate():
    H.bn_apply(y, scale, shift, None, True, z, M, Cp)
    return L.conv_fwd(z, geom, wf, stats=True, out=out)

def fused():
    part = L.empty((geom.nblk, 2, geom.out_cstride), torch.float32, y)
    H.igemm(y, wf, out, geom.M, geom.out_cstride, geom.Kf, geom.g_fwd(), geom.Kf, geom.out_cstride, b_rows=geom.Co,
            colstats=part, ldstat=geom.out_cstride, bna=(scale, shift, True))
    return out, part

dy = torch.randn(geom.M, geom.out_cstride, device="cuda").to(torch.bfloat16)

a, pa = separate(); a = a.clone(); pa = pa.clone()
b, pb = fused()
torch.cuda.synchronize()
print("max |fused - separate| =", (a.float() - b.float()).abs().max().item(), "of", a.float().abs().max().item(),
      "; stats", (pa - pb).abs().max().item())
for _ in range(3):
    t_bn = timeit(lambda: H.bn_apply(y, scale, shift, None, True, z, M, Cp))
    t_conv = timeit(lambda: L.conv_fwd(z, geom, wf, stats=True, out=out))
    t_sep = timeit(separate)
    t_f = timeit(fused)
    t_w = timeit(lambda: L.conv_wgrad_raw(z, dy, geom))
    t_wf = timeit(lambda: L.conv_wgrad_raw(y, dy, geom, x_bn=(scale, shift, True)))
    print(f"bn_apply {t_bn:.0f} us + temporal conv {t_conv:.0f} us = {t_sep:.0f} us together; fused {t_f:.0f} us | "
          f"weight gradient {t_w:.0f} us, with the fused apply {t_wf:.0f} us", flush=True)
