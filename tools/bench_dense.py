"""Dense GEMM micro-benchmark (pp_igemm dense mode / pp_wgrad): the wav2vec2 transformer shapes at M = 64 x 114 with the
epilogues the step uses, plus large square problems for the asymptotic rate of the GEMM core."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

dev = "cuda"
for opt in ("ring_igemm", "persistent_igemm", "xcd_remap_igemm"):
    if opt.upper() in os.environ:
        H.set_option(opt, int(os.environ[opt.upper()]))


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def case(M, N, K, n=20):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    wf, wt = L.prep_linear(w)
    fl = 2.0 * M * N * K
    out = []
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("bias+gelu", dict(bias=bias, act=H.ACT_GELU)),
                     ("bias+gelu+drop", dict(bias=bias, act=H.ACT_GELU, dropout=(0.1, 7))),
                     ("bias+drop+res", dict(bias=bias, dropout=(0.1, 7), residual=res))):
        t = timeit(lambda: L.linear_fwd(x, M, wf, N, **kw), n)
        out.append(f"{name} {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TF")
    t = timeit(lambda: L.linear_dgrad(dy, M, wt, K), n); out.append(f"dgrad {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TF")
    t = timeit(lambda: L.linear_wgrad(x, dy, M, N, K), n); out.append(f"wgrad {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TF")
    print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(out), flush=True)


M = 64 * 114
case(M, 2304, 768)
case(M, 768, 768)
case(M, 3072, 768)
case(M, 768, 3072)
case(64 * 229, 3072, 768)
case(8192, 8192, 8192, 5)
case(16384, 4096, 4096, 5)
case(8192, 1024, 8192, 5)
