"""Bandwidth of the BatchNorm passes at the layer-1 shapes of the C2 step (B=64): isolated, against the HBM roofline."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

dev = "cuda"
for opt in ("bn_nt", "bn_grid"):
    if opt.upper() in os.environ:
        H.set_option(opt, int(os.environ[opt.upper()]))
        print(opt, "=", os.environ[opt.upper()])


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


class BN:
    pass


for name, M, C in [("l1 mid 144", 3211264, 144), ("l1 out 64", 3211264, 64), ("l2 mid 288", 401408, 288), ("l2 out 128", 401408, 128)]:
    Cp = L.cpad(C)
    y = torch.randn(M, Cp, device=dev).to(torch.bfloat16)
    dz = torch.randn(M, Cp, device=dev).to(torch.bfloat16)
    bn = BN()
    bn.weight = torch.ones(C, device=dev); bn.bias = torch.zeros(C, device=dev)
    bn.running_mean = torch.zeros(C, device=dev); bn.running_var = torch.ones(C, device=dev)
    bn.num_batches_tracked = torch.zeros((), dtype=torch.long, device=dev)
    nblk = (M + 127) // 128
    partials = torch.zeros(nblk, 2, Cp, device=dev)
    H.colstats_bf16(y, M, Cp, partials, nblk)
    z, sv = L.bn_fwd(y, partials, nblk, M, bn, relu=True)
    elems = M * Cp
    t_apply = timeit(lambda: L.bn_fwd(y, partials, nblk, M, bn, relu=True, update_running=False))
    bnb = lambda: L.bn_bwd(dz, y, None, sv, bn.weight, relu=True)
    t_bwd = timeit(bnb)
    print(f"{name:12s} M={M} C={Cp}: fwd finalize+apply {t_apply*1e6:7.1f} us ({4*elems/t_apply/1e12:.2f} TB/s of y->z) | "
          f"bwd reduce+finalize+apply {t_bwd*1e6:7.1f} us ({10*elems/t_bwd/1e12:.2f} TB/s of 10 B/elem)", flush=True)
    # the backward passes one by one
    nb = min(2048, (M + 63) // 64)
    pb = torch.empty(nb, 2, Cp, device=dev)
    coef = torch.zeros(3, Cp, device=dev)
    dy = torch.empty_like(y)
    tr = timeit(lambda: H.bn_bwd_reduce(dz, y, None, sv.mean, sv.rstd, sv.scale, sv.shift, True, pb, nb, M, Cp))
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    tf = timeit(lambda: H.bn_bwd_finalize(pb, nb, sv.count, sv.C, Cp, bn.weight, sv.rstd, dg, db, coef))
    ta = timeit(lambda: H.bn_bwd_apply(dz, y, None, sv.mean, sv.rstd, coef, sv.scale, sv.shift, True, dy, None, M, Cp))
    print(f"             reduce {tr*1e6:7.1f} us ({4*elems/tr/1e12:.2f} TB/s) | finalize {tf*1e6:6.1f} us | apply {ta*1e6:7.1f} us ({6*elems/ta/1e12:.2f} TB/s)", flush=True)
