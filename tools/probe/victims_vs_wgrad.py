"""Is pp_layernorm_bwd the only kernel whose result changes beside the register-staged kernels?
Several small kernels of the audio tower on fixed inputs, each repeated while a second stream runs the dense weight gradient.
    python tools/probe/victims_vs_wgrad.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H
from peppa_amd import layers as L
from peppa_amd.hip import act16

H.set_deterministic(True)
H.set_option("ln_bwd_alone", 0)
dev = "cuda"
torch.manual_seed(0)
rows, D = 228, 768
ln = torch.nn.LayerNorm(D).to(dev)
x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
dy = (torch.randn(rows, D, device=dev) * 1e-4).to(torch.bfloat16)
_, saved = L.layernorm_fwd(x, ln)
u = torch.randn(rows, 3072, device=dev).to(torch.bfloat16)
dh = torch.randn(rows, 3072, device=dev).to(torch.bfloat16)
w = torch.randn(3072, 768, device=dev) * 0.05
wf, wt = L.prep_linear(w)
big = torch.randn(64 * 114, 768, device=dev).to(torch.bfloat16)
big_ln = L.layernorm_fwd(big, ln)[1]
dbig = (torch.randn(64 * 114, 768, device=dev) * 1e-4).to(torch.bfloat16)


def gelu_bwd():
    out = torch.empty_like(u)
    H.gelu_bwd(dh, u, out)
    return out


victims = {
    "layernorm_bwd 228 rows": lambda: L.layernorm_bwd(dy, x, ln, saved)[0],
    "layernorm_bwd 7296 rows": lambda: L.layernorm_bwd(dbig, big, ln, big_ln)[0],
    "layernorm_fwd 228 rows": lambda: L.layernorm_fwd(x, ln)[0],
    "layernorm_fwd 7296 rows": lambda: L.layernorm_fwd(big, ln)[0],
    "gelu_bwd 228 x 3072": gelu_bwd,
    "linear_fwd 228 x 768 -> 3072 (register-staged itself)": lambda: L.linear_fwd(x, rows, wf, 3072),
    "linear_dgrad 228 x 3072 -> 768": lambda: L.linear_dgrad(u, rows, wt, 768),
}
M = 64 * 114
xa = torch.randn(M, 768, device=dev).to(torch.bfloat16)
da = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
side = torch.cuda.Stream()
for name, fn in victims.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = torch.zeros((), device=dev, dtype=torch.int32)
    n = 0
    for it in range(150):
        with torch.cuda.stream(side):
            L.linear_wgrad(xa, da, M, 3072, 768)
        for _ in range(8):
            bad += (fn() != ref).any().to(torch.int32)
            n += 1
    torch.cuda.synchronize()
    print(f"{name:56s}: {int(bad.item())} of {n} launches differ", flush=True)
