import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H
dev = "cuda"
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 7296
for (N, K) in ((768, 768), (768, 3072), (2304, 768)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    gw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    print(f"N={N} K={K}: zeros {timeit(lambda: torch.zeros(N*K+N, device=dev)):.1f} us")
    for ms in (1, 2, 4, 8, 13, 16):
        t = timeit(lambda: H.wgrad(x, dy, gw, M, N, K, H.gather_dense(K), N, K, msplit=ms))
        tb = timeit(lambda: H.wgrad(x, dy, gw, M, N, K, H.gather_dense(K), N, K, msplit=ms, dbias=db))
        print(f"   msplit {ms:2d}: {t:7.1f} us ({2.0*M*N*K/t/1e6:6.1f} TF)   with bias {tb:7.1f} us")
