"""The stem over pixel pairs, two formulations of the same products (identical weight operand):
  A (shipped): taps (1,7,4) of 8 channels over [H][W/2][8], pad (0,3,2)        -> 28 gathers of 16 bytes per output position
  B: taps (1,7,1) of 32 channels over a row-padded [H][W/2 + 4][8] (two zero pairs in front), channel stride 8 < cg = 32
     -> 7 gathers of 64 contiguous bytes.
Forward and weight gradient, timings and equality.     python tools/probe/stem_wide_taps.py
(B needs pp_validate_gather's `cstride >= cg` check relaxed to `cstride > 0` in csrc/igemm.hip.  Measured round 4: forward
268 -> 236 us, weight gradient 337 -> 301 us, bit-equal forward -- 10-12 %, not worth a padded input layout: the generic kernels
are bound by their 16-byte-per-thread staging, not by the gather's line count.  profiles/r04_probe_stem_wide_taps.log)"""
import os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H, layers as L
dev = "cuda"
B, T, Hh, W2, Co = int(os.environ.get("B", "64")), 16, 112, 56, 45
Cop = 48


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, Hh, W2, 8, generator=g).to(torch.bfloat16).to(dev)             # pixel pairs
PADL, WP = 2, W2 + 4
xp = torch.zeros(B, T, Hh, WP, 8, dtype=torch.bfloat16, device=dev)
xp[:, :, :, PADL:PADL + W2] = x
w = (torch.randn(Cop, 7 * 4 * 8, generator=g) * 0.05).to(torch.bfloat16).to(dev)       # [Co][taps][8] = [Co][7][32]
Ho, Wo = 56, 56
M = B * T * Ho * Wo
dy = torch.randn(M, Cop, generator=g).to(torch.bfloat16).to(dev)
K = 224
gA = H.gather_conv(H.CONV_FWD, (T, Ho, Wo), (T, Hh, W2), (1, 7, 4), (1, 2, 1), (0, 3, 2), 8, 8)
gB = H.gather_conv(H.CONV_FWD, (T, Ho, Wo), (T, Hh, WP), (1, 7, 1), (1, 2, 1), (0, 3, 0), 32, 8)
outs = {}
for name, gg, xin in (("A taps (1,7,4) x 8ch ", gA, x), ("B taps (1,7,1) x 32ch", gB, xp)):
    y = torch.empty(M, Cop, dtype=torch.bfloat16, device=dev)
    st = torch.empty((M + 127) // 128, 2, Cop, device=dev)
    gw = torch.zeros(Cop, K, device=dev)
    fwd = lambda: H.igemm(xin.view(-1, 8), w, y, M, Cop, K, gg, K, Cop, b_rows=Cop, colstats=st, ldstat=Cop)
    def wg():
        gw.zero_()
        H.wgrad(xin.view(-1, 8), dy, gw, M, Cop, K, gg, Cop, K)
    tf, tw = timeit(fwd), timeit(wg)
    torch.cuda.synchronize()
    outs[name] = (y.float().clone(), gw.clone())
    print(f"{name}: forward {tf:7.1f} us   weight gradient {tw:7.1f} us", flush=True)
(ya, ga), (yb, gb) = outs.values()
print("forward equal:", torch.equal(ya, yb), " max |dW a - dW b| / max |dW| =", ((ga - gb).abs().max() / ga.abs().max()).item())
