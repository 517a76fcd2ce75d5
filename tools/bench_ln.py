"""Isolated timing of the LayerNorm / GELU / dropout passes at the transformer shape (M = 7296 rows)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

dev = "cuda"


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


class LN:
    pass


M = 7296
for D in (768, 512):
    x = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
    ln = LN(); ln.weight = torch.ones(D, device=dev); ln.bias = torch.zeros(D, device=dev); ln.eps = 1e-5
    y, saved = L.layernorm_fwd(x, ln)
    tf = timeit(lambda: L.layernorm_fwd(x, ln))
    tb = timeit(lambda: L.layernorm_bwd(dy, x, ln, saved))
    mb = M * D * 2 / 1e6
    print(f"LayerNorm D={D}: fwd {tf*1e6:6.1f} us ({2*mb/tf/1e6:.2f} TB/s) | bwd {tb*1e6:6.1f} us ({3*mb/tb/1e6:.2f} TB/s)")
u = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
du = torch.empty_like(u)
t = timeit(lambda: H.gelu_bwd(u, u, du))
print(f"gelu_bwd [M,3072]: {t*1e6:6.1f} us ({3*M*3072*2/t/1e12:.2f} TB/s)")
t = timeit(lambda: H.gelu_bwd_dropout(u, u, du, 0.1, 123))
print(f"gelu_bwd_dropout [M,3072]: {t*1e6:6.1f} us ({3*M*3072*2/t/1e12:.2f} TB/s)")
s = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: H.dropout_bf16(s, s, 0.1, 5))
print(f"dropout [M,768]: {t*1e6:6.1f} us ({2*M*768*2/t/1e12:.2f} TB/s)")
