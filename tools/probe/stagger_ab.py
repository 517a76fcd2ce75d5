"""A/B of the window kernels' staggered halves (pp_set_option("win_stagger", 0 / 1)) inside one process, alternating, on the
spatial (1,3,3) convolutions of the step; outputs and BatchNorm statistics of the two forms must be BIT-IDENTICAL (same
products, same accumulation order), checked on every repetition (a read that overtakes its DMA shows up as a mismatch)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

dev = "cuda"
B = int(os.environ.get("B", "64"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, Ci, Co, thw):
    geom = L.ConvGeom(B, thw, Ci, Co, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = torch.randn(geom.Min, geom.in_cstride, device=dev).to(torch.bfloat16)
    dy = torch.randn(geom.M, geom.out_cstride, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, Ci, 1, 3, 3, device=dev) * 0.05
    wf, wd = L.prep_conv_weights(w, geom)
    fl = 2.0 * geom.M * Co * 9 * Ci
    res = {}
    ref = {}
    for mode in (0, 1):
        H.set_option("win_stagger", mode)
        out = L.conv_fwd(x, geom, wf, stats=True)
        dx = L.conv_dgrad(dy, geom, wd)
        torch.cuda.synchronize()
        ref[mode] = ([t.clone() for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)], dx.clone())
    same = all(torch.equal(a, b) for a, b in zip(ref[0][0], ref[1][0])) and torch.equal(ref[0][1], ref[1][1])
    bad = 0
    for r in range(ROUNDS):
        for mode in (0, 1):
            H.set_option("win_stagger", mode)
            res.setdefault(("fwd", mode), []).append(timeit(lambda: L.conv_fwd(x, geom, wf, stats=True)))
            res.setdefault(("dgrad", mode), []).append(timeit(lambda: L.conv_dgrad(dy, geom, wd)))
            out = L.conv_fwd(x, geom, wf, stats=True)
            dx = L.conv_dgrad(dy, geom, wd)
            torch.cuda.synchronize()
            outs = [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)]
            if not (all(torch.equal(a, b) for a, b in zip(outs, ref[0][0])) and torch.equal(dx, ref[0][1])):
                bad += 1
    line = f"{name:24s} M={geom.M:8d} identical={same} mismatching repetitions={bad}"
    for op in ("fwd", "dgrad"):
        a, b = res[(op, 0)], res[(op, 1)]
        line += f" | {op}: lockstep {min(a):7.1f}/{sorted(a)[len(a) // 2]:7.1f} us, staggered {min(b):7.1f}/{sorted(b)[len(b) // 2]:7.1f} us ({fl / min(b) / 1e6:5.0f} TF)"
    print(line, flush=True)


case("l1 spatial 64->144", 64, 144, (16, 56, 56))
case("l1 spatial 144->64 (dgrad shape)", 144, 64, (16, 56, 56))
case("l2 spatial 128->288", 128, 288, (8, 28, 28))
case("l2.0 spatial 128->230", 128, 230, (8, 28, 28))
case("l3 spatial 256->576", 256, 576, (4, 14, 14))
case("l4 spatial 512->1152", 512, 1152, (2, 7, 7))
H.set_option("win_stagger", 0)
