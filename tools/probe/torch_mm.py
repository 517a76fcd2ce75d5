"""What the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) does on the transformer shapes, next to tools/bench_dense.py."""
import torch

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for M, N, K in ((7296, 2304, 768), (7296, 768, 768), (7296, 3072, 768), (7296, 768, 3072), (14656, 3072, 768), (8192, 8192, 8192), (235456, 512, 1536)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: torch.mm(x, w.t()))          # forward  y = x W^T
    t2 = timeit(lambda: torch.mm(dy, w))             # dgrad    dx = dy W
    t3 = timeit(lambda: torch.mm(dy.t(), x))         # wgrad    dW = dy^T x   (bf16 out)
    t4 = timeit(lambda: torch.mm(dy.t().float(), x.float())) if M * N < 3e7 else float("nan")
    print(f"M={M:6d} N={N:5d} K={K:5d}: fwd {t1*1e6:7.1f} us {fl/t1/1e12:6.0f} TF | dgrad {t2*1e6:7.1f} us {fl/t2/1e12:6.0f} TF | "
          f"wgrad(bf16 out) {t3*1e6:7.1f} us {fl/t3/1e12:6.0f} TF", flush=True)
