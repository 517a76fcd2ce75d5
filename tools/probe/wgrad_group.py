"""What would a grouped, unsplit launch of the transformer's weight gradients reach with the existing kernel cores?
nbatch = 12 problems (one per layer) of M = 7296 rows each, msplit = 1 (every tile reduces its whole M: no atomics from
splitting) against the shipped per-layer launches (automatic split).  Also the LDS-DMA ring core on one stacked problem."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H
dev = "cuda"


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, NB = 7296, 12
for (N, K) in ((768, 768), (768, 3072), (3072, 768), (2304, 768)):
    x = torch.randn(NB, M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(NB, M, N, device=dev).to(torch.bfloat16)
    gw = torch.zeros(NB, N, K, device=dev)
    db = torch.zeros(NB, N, device=dev)
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: [H.wgrad(x[i], dy[i], gw[i], M, N, K, H.gather_dense(K), N, K, dbias=db[i]) for i in range(NB)])
    line = f"N={N:5d} K={K:5d}: 12 launches (auto split, bias) {t1:8.1f} us {NB * fl / t1 / 1e6:6.0f} TF"
    for ms in (1, 2, 3):
        t = timeit(lambda: H.wgrad(x, dy, gw, M, N, K, H.gather_dense(K), N, K, msplit=ms, nbatch=NB, x_s=M * K, dy_s=M * N, dw_s=N * K,
                                   dbias=db, dbias_s=N))
        line += f" | grouped msplit {ms}: {t:8.1f} us {NB * fl / t / 1e6:6.0f} TF"
    print(line, flush=True)
    # one stacked problem of 12 M rows: the same FLOPs through the ring core (nbatch must be 1 there)
    xs, dys = x.view(NB * M, K), dy.view(NB * M, N)
    for ring in (0, 1):
        H.set_option("ring_wgrad", ring)
        for ms in (0, 1, 2):
            t = timeit(lambda: H.wgrad(xs, dys, gw[0], NB * M, N, K, H.gather_dense(K), N, K, msplit=ms))
            print(f"      stacked M={NB * M} ring={ring} msplit={ms}: {t:8.1f} us {NB * fl / t / 1e6:6.0f} TF", flush=True)
    H.set_option("ring_wgrad", 0)
