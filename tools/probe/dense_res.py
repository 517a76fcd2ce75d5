"""Transformer GEMMs with fused epilogues at M = 7296 (bias + GELU + pre-activation copy; data gradients with residual):
    python tools/probe/dense_res.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import layers as L

dev = "cuda"
M = 64 * 114


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = []
for name, N, K in (("qkv", 2304, 768), ("out", 768, 768), ("ffn1", 3072, 768), ("ffn2", 768, 3072)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    res = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    wf, wt = L.prep_linear(w)
    t_f = timeit(lambda: L.linear_fwd(x, M, wf, N, bias=b))
    t_d = timeit(lambda: L.linear_dgrad(dy, M, wt, K))
    t_r = timeit(lambda: L.linear_dgrad(dy, M, wt, K, residual=res))
    out.append(f"{name}: fwd+bias {t_f:6.1f} us, dgrad {t_d:6.1f} us, dgrad+residual {t_r:6.1f} us")
print(" | ".join(out))
