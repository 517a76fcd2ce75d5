"""Reference point: what does the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reach on the transformer's plain GEMM shapes?
Forward X W^T, data gradient dY W, weight gradient dY^T X (bf16 in, fp32 accumulate; outputs bf16 / fp32 as the library's).
    python tools/probe/vendor_gemm.py"""
import os
import sys
import warnings

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = 64 * 114
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    t_f = timeit(lambda: torch.mm(x, w.t(), out=y))
    t_d = timeit(lambda: torch.mm(dy, w, out=dx))
    t_w = timeit(lambda: torch.mm(dy.t(), x, out=dw))
    xs, dys = x.unsqueeze(0).expand(12, M, K).contiguous(), dy.unsqueeze(0).expand(12, M, N).contiguous()
    dws = torch.empty(12, N, K, device="cuda", dtype=torch.bfloat16)
    t_wb = timeit(lambda: torch.bmm(dys.transpose(1, 2), xs, out=dws))
    print(f"N={N:5d} K={K:5d}:  fwd {t_f*1e6:7.1f} us {fl/t_f/1e12:6.0f} TF | dgrad {t_d*1e6:7.1f} us {fl/t_d/1e12:6.0f} TF | "
          f"wgrad {t_w*1e6:7.1f} us {fl/t_w/1e12:6.0f} TF | wgrad x12 (bmm) {t_wb*1e6:7.1f} us {12*fl/t_wb/1e12:6.0f} TF", flush=True)
# the audio feature extractor's conv1 as the dense GEMM it is (rows overlap: lda = 1024, K = 1536)
T1 = 7359
To = (T1 - 3) // 2 + 1
xf = torch.randn(64 * T1 * 512 + 4096, device="cuda").bfloat16()
a = torch.as_strided(xf, (64 * To, 1536), (1024, 1))          # (clip boundaries ignored: a timing stand-in)
w = torch.randn(512, 1536, device="cuda").bfloat16()
y = torch.empty(64 * To, 512, device="cuda", dtype=torch.bfloat16)
try:
    t = timeit(lambda: torch.mm(a, w.t(), out=y))
    print(f"conv1 as GEMM M={64*To} N=512 K=1536 (overlapping rows): {t*1e6:7.1f} us {2.0*64*To*512*1536/t/1e12:6.0f} TF")
except Exception as e:
    print("conv1 as strided GEMM:", type(e).__name__, str(e)[:100])
ac = a.contiguous()
t = timeit(lambda: torch.mm(ac, w.t(), out=y))
print(f"conv1 as GEMM, contiguous rows: {t*1e6:7.1f} us {2.0*64*To*512*1536/t/1e12:6.0f} TF")
