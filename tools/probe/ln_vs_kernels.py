"""Which concurrent kernel makes pp_layernorm_bwd (fixed inputs) return a different dx?  One stream repeats the LayerNorm
backward, a second stream runs ONE kind of kernel of the video trunk / audio tower at a time.
    python tools/probe/ln_vs_kernels.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peppa_amd import hip as H
from peppa_amd import layers as L

H.set_deterministic(True)
if "LN_BWD_ALONE" in os.environ:
    H.set_option("ln_bwd_alone", int(os.environ["LN_BWD_ALONE"]))
dev = "cuda"
torch.manual_seed(0)
rows, D = int(os.environ.get('ROWS', '228')), int(os.environ.get('D', '768'))
ln = torch.nn.LayerNorm(D).to(dev)
x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
dy = (torch.randn(rows, D, device=dev) * 1e-4).to(torch.bfloat16)
y, saved = L.layernorm_fwd(x, ln)
ref = L.layernorm_bwd(dy, x, ln, saved)[0].clone()
side = torch.cuda.Stream()
B = int(os.environ.get("B", "16"))


def conv(Ci, Co, k, s, p, thw):
    geom = L.ConvGeom(B, thw, Ci, Co, k, s, p)
    xx = torch.randn(geom.Min, geom.in_cstride, device=dev).to(torch.bfloat16)
    dd = torch.randn(geom.M, geom.out_cstride, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, Ci, *k, device=dev) * 0.05
    wf, wd = L.prep_conv_weights(w, geom)
    return geom, xx, dd, wf, wd


cases = {}
g1 = conv(64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), (16, 56, 56))
cases["window fwd 64->144"] = lambda: L.conv_fwd(g1[1], g1[0], g1[3], stats=True)
cases["window dgrad 144->64"] = lambda: L.conv_dgrad(g1[2], g1[0], g1[4])
cases["wgrad_sw 64->144"] = lambda: L.conv_wgrad_raw(g1[1], g1[2], g1[0])
g2 = conv(144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (16, 56, 56))
cases["temporal window fwd 144->64"] = lambda: L.conv_fwd(g2[1], g2[0], g2[3], stats=True)
cases["temporal window dgrad"] = lambda: L.conv_dgrad(g2[2], g2[0], g2[4])
cases["wgrad_tw 144->64"] = lambda: L.conv_wgrad_raw(g2[1], g2[2], g2[0])
g3 = conv(64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (16, 56, 56))
cases["generic fwd 64->230 s2"] = lambda: L.conv_fwd(g3[1], g3[0], g3[3], stats=True)
cases["generic dgrad s2"] = lambda: L.conv_dgrad(g3[2], g3[0], g3[4])
cases["generic wgrad s2"] = lambda: L.conv_wgrad_raw(g3[1], g3[2], g3[0])
M = 64 * 114
xa = torch.randn(M, 768, device=dev).to(torch.bfloat16)
wa = torch.randn(3072, 768, device=dev) * 0.05
wfa, wta = L.prep_linear(wa)
da = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
cases["dense ring fwd 768->3072"] = lambda: L.linear_fwd(xa, M, wfa, 3072)
cases["dense ring dgrad"] = lambda: L.linear_dgrad(da, M, wta, 768)
cases["dense wgrad"] = lambda: L.linear_wgrad(xa, da, M, 3072, 768)
big = torch.randn(B * 16 * 56 * 56, 64, device=dev).to(torch.bfloat16)
cases["elementwise (torch relu)"] = lambda: torch.relu(big)
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
cases["torch.mm"] = lambda: torch.mm(a, a)
only = os.environ.get("CASE", "")
for name, fn in cases.items():
    if only and only not in name:
        continue
    fn()
    torch.cuda.synchronize()
    bad = torch.zeros((), device=dev, dtype=torch.int32)
    n = 0
    for it in range(int(os.environ.get("ITERS", "150"))):
        with torch.cuda.stream(side):
            fn()
        for _ in range(8):
            out = L.layernorm_bwd(dy, x, ln, saved)[0]
            bad += (out != ref).any().to(torch.int32)
            n += 1
    torch.cuda.synchronize()
    print(f"{name:32s}: {int(bad.item())} of {n} LayerNorm-backward launches differ", flush=True)

if os.environ.get("DETAIL"):
    name = os.environ["DETAIL"]
    fn = cases[name]
    shown = 0
    fails = []
    gam0, x0, dy0, m0, r0 = ln.weight.detach().clone(), x.clone(), dy.clone(), saved[0].clone(), saved[1].clone()
    for it in range(400):
        with torch.cuda.stream(side):
            fn()
        outs = [L.layernorm_bwd(dy, x, ln, saved) for _ in range(8)]
        for o in outs:
            d = (o[0].float() - ref.float())
            if (d != 0).any().item():
                nz = (d != 0).nonzero()
                rws = sorted(set(nz[:, 0].tolist()))
                r = rws[0]
                cols = nz[nz[:, 0] == r][:, 1].tolist()
                print(f"iteration {it}: rows {rws}; row {r}: {len(cols)} columns differ, first {cols[:12]}; ref {ref[r, cols[:6]].tolist()} got {o[0][r, cols[:6]].tolist()}", flush=True)
                shown += 1
                fails.append((rws, o[0][rws].cpu().clone()))
        if shown >= 12:
            break
    torch.cuda.synchronize()
    same = [torch.equal(a, b) for a, b in ((gam0, ln.weight), (x0, x), (dy0, dy), (m0, saved[0]), (r0, saved[1]))]
    os.makedirs("gpurun_out/r3", exist_ok=True)
    torch.save({"x": x.cpu(), "dy": dy.cpu(), "gamma": ln.weight.detach().cpu(), "mean": saved[0].cpu(), "rstd": saved[1].cpu(),
                "ref": ref.cpu(), "fails": fails}, "gpurun_out/r3/ln_fail.pt")
    print("inputs unchanged at the end (gamma, x, dy, mean, rstd):", same)
