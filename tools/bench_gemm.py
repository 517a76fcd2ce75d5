"""Micro-benchmark of the implicit-GEMM / wgrad kernels on the hot shapes of the C2 step (B=64)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L

dev = "cuda"
if os.environ.get("MASKED_STRIDED_DGRAD"):
    L.MASKED_STRIDED_DGRAD = True
B = int(os.environ.get("B", "64"))
only = sys.argv[1] if len(sys.argv) > 1 else ""
case_filter = os.environ.get("CASE", "")
for opt in ("xcd_remap_igemm", "xcd_remap_wgrad", "persistent_igemm", "ring_igemm", "ring_wgrad", "sw_wgrad", "win_igemm", "win_tall", "win_temporal", "win_out_nt", "ring_wn", "win_producers", "tw_producers", "tw_narrow", "ring_producers", "win_kpb", "win_s2d", "wgrad_flat", "wgrad_big"):
    if opt.upper() in os.environ:
        H.set_option(opt, int(os.environ[opt.upper()]))


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def conv_case(name, Ci, Co, k, s, p, thw, in_cstride=None, out_cstride=None):
    if case_filter and case_filter not in name:
        return
    geom = L.ConvGeom(B, thw, Ci, Co, k, s, p, in_cstride=in_cstride, out_cstride=out_cstride)
    x = torch.randn(geom.Min, geom.in_cstride, device=dev).to(torch.bfloat16)
    dy = torch.randn(geom.M, geom.out_cstride, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, Ci, *k, device=dev) * 0.05
    wf, wd = L.prep_conv_weights(w, geom)
    fl = 2.0 * geom.M * Co * geom.taps * Ci
    res = []
    if not only or "fwd" in only:
        t = timeit(lambda: L.conv_fwd(x, geom, wf, stats=True)); res.append(f"fwd {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    if not only or "dgrad" in only:
        t = timeit(lambda: L.conv_dgrad(dy, geom, wd)); res.append(f"dgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    if not only or "wgrad" in only:
        t = timeit(lambda: L.conv_wgrad_raw(x, dy, geom)); res.append(f"wgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    print(f"{name:28s} M={geom.M:8d} " + " | ".join(res), flush=True)


def dense_case(name, M, N, K):
    if case_filter and case_filter not in name:
        return
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * 0.05
    wf, wt = L.prep_linear(w)
    fl = 2.0 * M * N * K
    res = []
    if not only or "fwd" in only:
        t = timeit(lambda: L.linear_fwd(x, M, wf, N)); res.append(f"fwd {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    if not only or "dgrad" in only:
        t = timeit(lambda: L.linear_dgrad(dy, M, wt, K)); res.append(f"dgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    if not only or "wgrad" in only:
        t = timeit(lambda: L.linear_wgrad(x, dy, M, N, K, want_bias=not os.environ.get('NO_BIAS'))); res.append(f"wgrad {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    print(f"{name:28s} M={M:8d} " + " | ".join(res), flush=True)


# layer-1 geometry: GEOM="T,H,W" (default 16,56,56 = BASELINE configs[1]; the reference's own clips: GEOM=23,50,90 B=8)
T, S1, S2 = (int(v) for v in os.environ.get("GEOM", "16,56,56").split(","))
half = lambda v: (v + 1) // 2
g1 = (T, S1, S2)
g2 = (half(T), half(S1), half(S2))
g3 = tuple(half(v) for v in g2)
g4 = tuple(half(v) for v in g3)
conv_case("stem2 45->64 (3,1,1)", 45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), g1)
conv_case("l1 spatial 64->144", 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), g1)
conv_case("l1 temporal 144->64", 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), g1)
conv_case("l2.0 spatial 64->230 s2", 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), g1)
conv_case("l2.0 temporal 230->128 s2", 230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (T, g2[1], g2[2]))
conv_case("l2 spatial 128->288", 128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), g2)
conv_case("l2 temporal 288->128", 288, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), g2)
conv_case("l3 spatial 256->576", 256, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1), g3)
conv_case("l3 temporal 576->256", 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), g3)
conv_case("l4 spatial 512->1152", 512, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1), g4)
conv_case("l4 temporal 1152->512", 1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), g4)
conv_case("audio conv1 k3s2 T=7359", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (7359, 1, 1))
# the same with rows of 576 / 640 instead of 512 channels (1152 / 1280 B instead of a power of two: HBM channel spread)
conv_case("audio conv1 rows of 576", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (7359, 1, 1), in_cstride=576, out_cstride=576)
conv_case("audio conv1 rows of 520", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (7359, 1, 1), in_cstride=520, out_cstride=520)
conv_case("audio conv2 k3s2 T=3679", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (3679, 1, 1))
conv_case("audio conv2 rows of 576", 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (3679, 1, 1), in_cstride=576, out_cstride=576)
conv_case("audio conv5 k2s2 T=459", 512, 512, (2, 1, 1), (2, 1, 1), (0, 0, 0), (459, 1, 1))
conv_case("audio conv5 rows of 576", 512, 512, (2, 1, 1), (2, 1, 1), (0, 0, 0), (459, 1, 1), in_cstride=576, out_cstride=576)
M = B * int(os.environ.get("AUDIO_T", "114"))
dense_case("qkv 768->2304", M, 2304, 768)
dense_case("out 768->768", M, 768, 768)
dense_case("ffn1 768->3072", M, 3072, 768)
dense_case("ffn2 3072->768", M, 768, 3072)
