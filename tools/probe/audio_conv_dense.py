"""The wav2vec2 feature-extractor convs (512 -> 512, k 3 / 2, stride 2) are dense GEMMs with overlapping rows: output row t
reads input rows 2 t .. 2 t + k - 1, which are CONTIGUOUS in the channels-last [T][512] tensor: A[m] = x + m * 1024,
K = k * 512.  Conv-gather path (what the step runs) against the dense path with lda = 1024 on the same data."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from peppa_amd import hip as H, layers as L
dev = "cuda"


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 64
for (Tin, k) in ((7359, 3), (3679, 3), (1839, 3), (919, 3), (459, 2), (229, 2)):
    geom = L.ConvGeom(B, (Tin, 1, 1), 512, 512, (k, 1, 1), (2, 1, 1), (0, 0, 0))
    x = torch.randn(geom.Min, 512, device=dev).to(torch.bfloat16)
    w = torch.randn(512, 512, k, 1, 1, device=dev) * 0.03
    wf, wd = L.prep_conv_weights(w, geom)
    fl = 2.0 * geom.M * 512 * k * 512
    pre = torch.empty(geom.M, 512, device=dev, dtype=torch.bfloat16)
    t_conv = timeit(lambda: L.conv_fwd(x, geom, wf, act=H.ACT_GELU, pre=pre))
    y1, _ = L.conv_fwd(x, geom, wf, act=H.ACT_GELU, pre=pre)
    # dense: one clip at a time would need per-clip bases (rows of different clips are not 1024 apart when Tin is odd):
    # batch the clips with nbatch = B, a_s = Tin * 512, c_s = To * 512
    To = geom.To
    y2 = torch.empty(geom.M, 512, device=dev, dtype=torch.bfloat16)
    pre2 = torch.empty_like(y2)

    def dense():
        H.igemm(x, wf, y2, To, 512, k * 512, H.gather_dense(1024), k * 512, 512, b_rows=512, act=H.ACT_GELU, Cpre=pre2,
                nbatch=B, inner=1, a_s=(Tin * 512, 0), b_s=(0, 0), c_s=(To * 512, 0))
    try:
        t_dense = timeit(dense)
        dense(); torch.cuda.synchronize()
        same = torch.equal(y1, y2)
        err = (y1.float() - y2.float()).abs().max().item()
    except Exception as e:
        t_dense, same, err = float("nan"), str(e)[:80], 0
    print(f"T={Tin:5d} k={k} M={geom.M:7d}: conv path {t_conv:7.1f} us {fl / t_conv / 1e6:5.0f} TF | dense lda=1024, nbatch={B}: {t_dense:7.1f} us "
          f"{fl / t_dense / 1e6:5.0f} TF | same {same} max diff {err:.3g}", flush=True)
