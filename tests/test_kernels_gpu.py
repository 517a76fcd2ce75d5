"""GPU parity of every HIP kernel (through the C ABI) against plain fp32 PyTorch on the CPU.

Inputs are rounded to bf16 first so both sides see identical operands; tolerances below are
for bf16 outputs of fp32-accumulated contractions (rel 2^-8 per rounding)."""
import math
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from peppa_amd import hip as H
from peppa_amd import layers as L

DEV = "cuda"


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_cl(x, cp):
    """(B,C,T,H,W) fp32 -> channels-last bf16 [B*T*H*W][cp] on the GPU."""
    B, C = x.shape[:2]
    y = x.permute(0, 2, 3, 4, 1).reshape(-1, C)
    out = torch.zeros(y.shape[0], cp)
    out[:, :C] = y
    return out.to(torch.bfloat16).to(DEV)


def from_cl(y, B, thw, C):
    return y.float().cpu()[:, :C].reshape(B, *thw, C).permute(0, 4, 1, 2, 3)


def close(a, b, rtol=2e-2, atol=None, name=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = b.abs().max().item() + 1e-12
    atol = atol if atol is not None else 1e-2 * scale
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{name}: non-finite output"
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * scale, f"{name}: max err {err:.4g} vs scale {scale:.4g}"
    if scale > 1e-6:  # cosine is meaningless for an (analytically) zero reference such as db2
        cos = F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
        assert cos > 0.9995, f"{name}: cosine {cos}"


CONV_CASES = [
    # Ci, Co, k, s, p, B, T, H, W
    (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 3, 10, 12),
    (144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 5, 6, 7),
    (64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), 2, 3, 10, 12),
    (230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), 2, 6, 5, 5),
    (64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0), 2, 4, 8, 8),
    (45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 1, 4, 9, 9),
    (32, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 4, 6, 6),
    (32, 40, (3, 3, 3), (2, 2, 2), (1, 1, 1), 1, 5, 7, 9),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(case):
    Ci, Co, k, s, p, B, T, Hh, W = case
    g = torch.Generator().manual_seed(Ci * 7 + Co)
    x = rb(torch.randn(B, Ci, T, Hh, W, generator=g))
    w = rb(torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * k[0] * k[1] * k[2]))
    x.requires_grad_(); w.requires_grad_()
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    xc = to_cl(x.detach(), geom.in_cstride)
    wf, wd = L.prep_conv_weights(w.detach().to(DEV).contiguous(), geom)
    y, partials = L.conv_fwd(xc, geom, wf, stats=True)
    torch.cuda.synchronize()
    close(from_cl(y, B, geom.out_thw, Co), y_ref, name="conv fwd")
    assert (y.float().cpu()[:, Co:] == 0).all(), "padded output channels must be zero"
    yf = y.float().cpu()
    ps = partials.cpu().sum(0)
    close(ps[0][:Co], yf.sum(0)[:Co], atol=2e-2 * yf.abs().sum(0).max().item(), name="colstats sum")
    close(ps[1][:Co], (yf * yf).sum(0)[:Co], name="colstats sumsq")

    dyc = to_cl(dy, geom.out_cstride)
    dx = L.conv_dgrad(dyc, geom, wd)
    dw = L.conv_wgrad(xc, dyc, geom, w.shape)
    torch.cuda.synchronize()
    close(from_cl(dx, B, (T, Hh, W), Ci), x.grad, name="conv dgrad")
    close(dw, w.grad, name="conv wgrad")


def test_stem_conv_cin3():
    g = torch.Generator().manual_seed(1)
    B, T, Hh, W = 2, 3, 20, 24
    x = torch.rand(B, 3, T, Hh, W, generator=g)
    mean, std = (0.62745821, 0.66273642, 0.66865104), (0.24167268, 0.20884572, 0.27490067)
    xn = (x - torch.tensor(mean).view(1, 3, 1, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1, 1)
    w = rb(torch.randn(45, 3, 1, 7, 7, generator=g) / 12).requires_grad_()
    xr = rb(xn)
    y_ref = F.conv3d(xr, w, stride=(1, 2, 2), padding=(0, 3, 3))
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    geom = L.ConvGeom(B, (T, Hh, W), 3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), in_cstride=8, cg_in=8)
    xc = torch.empty(B * T * Hh * W, 8, dtype=torch.bfloat16, device=DEV)
    H.video_normalize_ndhwc(x.to(DEV), xc, mean, std)
    torch.cuda.synchronize()
    close(xc.float().cpu()[:, :3].reshape(B, T, Hh, W, 3).permute(0, 4, 1, 2, 3), xn, name="video normalize")
    assert (xc.float().cpu()[:, 3:] == 0).all()
    wf, _ = L.prep_conv_weights(w.detach().to(DEV), geom, need_dgrad=False)
    y, _ = L.conv_fwd(xc, geom, wf, stats=True)
    dw = L.conv_wgrad(xc, to_cl(dy, geom.out_cstride), geom, w.shape)
    torch.cuda.synchronize()
    close(from_cl(y, B, geom.out_thw, 45), y_ref, name="stem fwd")
    close(dw, w.grad, name="stem wgrad")


def test_basic_stem_conv_3x7x7_cin3():
    """r3d_18 / mc3_18 `BasicStem`: Conv3d(3, 64, (3,7,7), stride (1,2,2), pad (1,3,3)) -- 147 taps (the tap tables of
    pp_igemm / pp_wgrad held 128 in round 1, so these two backbones could not run at all)."""
    g = torch.Generator().manual_seed(11)
    B, T, Hh, W = 2, 4, 18, 22
    x = rb(torch.randn(B, 3, T, Hh, W, generator=g))
    w = rb(torch.randn(64, 3, 3, 7, 7, generator=g) / 21).requires_grad_()
    y_ref = F.conv3d(x, w, stride=(1, 2, 2), padding=(1, 3, 3))
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    geom = L.ConvGeom(B, (T, Hh, W), 3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), in_cstride=8, cg_in=8)
    xc = to_cl(x, 8)
    wf, _ = L.prep_conv_weights(w.detach().to(DEV), geom, need_dgrad=False)
    y, _ = L.conv_fwd(xc, geom, wf, stats=True)
    dw = L.conv_wgrad(xc, to_cl(dy, geom.out_cstride), geom, w.shape)
    torch.cuda.synchronize()
    close(from_cl(y, B, geom.out_thw, 64), y_ref, name="basic stem fwd")
    close(dw, w.grad, name="basic stem wgrad")
    with pytest.raises(Exception, match="taps"):     # 3 x 9 x 11 = 297 taps: rejected, not silently wrong
        bad = L.ConvGeom(1, (3, 12, 12), 3, 16, (3, 9, 11), (1, 1, 1), (1, 4, 5), in_cstride=8, cg_in=8)
        L.conv_fwd(to_cl(torch.zeros(1, 3, 3, 12, 12), 8), bad, torch.zeros(16, 297, 8, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(Exception, match="taps"):
        L.conv_wgrad_raw(to_cl(torch.zeros(1, 3, 3, 12, 12), 8), torch.zeros(432, 16, dtype=torch.bfloat16, device=DEV), bad)


@pytest.mark.parametrize("k,s,T", [(3, 2, 41), (2, 2, 30)])
def test_conv1d_stack_layer(k, s, T):
    """wav2vec2 feature-extractor convs: Conv1d(512,512,k,s) as a strided GEMM + GELU epilogue."""
    g = torch.Generator().manual_seed(k)
    B, C = 3, 512
    x = rb(torch.randn(B, C, T, generator=g)).requires_grad_()
    w = rb(torch.randn(C, C, k, generator=g) / math.sqrt(C * k)).requires_grad_()
    u_ref = F.conv1d(x, w, stride=s)
    y_ref = F.gelu(u_ref)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    geom = L.ConvGeom(B, (T, 1, 1), C, C, (k, 1, 1), (s, 1, 1), (0, 0, 0))
    xc = x.detach().permute(0, 2, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous()
    wf, wd = L.prep_conv_weights(w.detach().to(DEV), geom)
    pre = torch.empty(geom.M, C, dtype=torch.bfloat16, device=DEV)
    y, _ = L.conv_fwd(xc, geom, wf, act=H.ACT_GELU, pre=pre)
    torch.cuda.synchronize()
    To = geom.To
    close(y.float().cpu().reshape(B, To, C).permute(0, 2, 1), y_ref, name="conv1d+gelu")
    close(pre.float().cpu().reshape(B, To, C).permute(0, 2, 1), u_ref, name="conv1d pre-activation")
    dyc = dy.permute(0, 2, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous()
    du = torch.empty_like(dyc)
    H.gelu_bwd(dyc, pre, du)
    dx = L.conv_dgrad(du, geom, wd)
    dw = L.conv_wgrad(xc, du, geom, w.shape)
    torch.cuda.synchronize()
    close(dx.float().cpu().reshape(B, T, C).permute(0, 2, 1), x.grad, name="conv1d dgrad")
    close(dw, w.grad, name="conv1d wgrad")


def test_grouped_posconv():
    g = torch.Generator().manual_seed(9)
    B, T, C, G, K = 2, 37, 768, 16, 128
    x = rb(torch.randn(B, C, T, generator=g)).requires_grad_()
    w = rb(torch.randn(C, C // G, K, generator=g) / math.sqrt(K * C // G)).requires_grad_()
    bias = torch.randn(C, generator=g)
    y_ref = F.conv1d(x, w, bias, padding=K // 2, groups=G)[..., :-1]
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    geom = L.ConvGeom(B, (T, 1, 1), C, C, (K, 1, 1), (1, 1, 1), (K // 2, 0, 0), groups=G, To=T)
    xc = x.detach().permute(0, 2, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous()
    wf, wd = L.prep_conv_weights(w.detach().to(DEV), geom)
    y, _ = L.conv_fwd(xc, geom, wf, bias=bias.to(DEV))
    dyc = dy.permute(0, 2, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous()
    dx = L.conv_dgrad(dyc, geom, wd)
    dw = L.conv_wgrad(xc, dyc, geom, w.shape)
    torch.cuda.synchronize()
    close(y.float().cpu().reshape(B, T, C).permute(0, 2, 1), y_ref, name="posconv fwd")
    close(dx.float().cpu().reshape(B, T, C).permute(0, 2, 1), x.grad, name="posconv dgrad")
    close(dw, w.grad, name="posconv wgrad")


@pytest.mark.parametrize("M,N,K", [(300, 768, 512), (257, 28, 768), (64, 3072, 768), (1000, 768, 3072), (5, 512, 28)])
def test_linear_fwd_bwd(M, N, K):
    g = torch.Generator().manual_seed(M + N)
    x = rb(torch.randn(M, K, generator=g)).requires_grad_()
    w = rb(torch.randn(N, K, generator=g) / math.sqrt(K)).requires_grad_()
    b = torch.randn(N, generator=g).requires_grad_()
    res = rb(torch.randn(M, N, generator=g))
    y_ref = F.linear(x, w, b) + res
    dy = rb(torch.randn(M, N, generator=g))
    y_ref.backward(dy)
    Kp, Np = L.cpad(K), L.cpad(N)

    def padded(t, cols):
        o = torch.zeros(t.shape[0], cols)
        o[:, :t.shape[1]] = t
        return o.to(torch.bfloat16).to(DEV)
    xc, rc, dyc = padded(x.detach(), Kp), padded(res, Np), padded(dy, Np)
    wf, wt = L.prep_linear(w.detach().to(DEV))
    bd = b.detach().to(DEV)
    y = L.linear_fwd(xc, M, wf, N, bias=bd, residual=rc)
    dx = L.linear_dgrad(dyc, M, wt, K)
    dw, db = L.linear_wgrad(xc, dyc, M, N, K)
    torch.cuda.synchronize()
    close(y[:, :N], y_ref, name="linear fwd")
    close(dx[:, :K], x.grad, name="linear dgrad")
    close(dw, w.grad, name="linear wgrad")
    close(db, b.grad, name="linear bias grad")
    y32 = L.linear_fwd(xc, M, wf, N, bias=bd, out_f32=True)
    torch.cuda.synchronize()
    close(y32[:, :N], y_ref - res, rtol=2e-3, atol=2e-3 * y_ref.abs().max().item(), name="linear fp32 out")


@pytest.mark.parametrize("M,N,K,n", [(7296, 768, 768, 5), (1000, 768, 3072, 3), (300, 3072, 768, 2), (257, 2304, 768, 4), (70, 28, 768, 3)])
def test_grouped_linear_wgrad_matches_torch_and_is_bitwise_reproducible(M, N, K, n):
    """the same Linear shape of several transformer layers as ONE launch (pp_wgrad_desc.ptr_table):
    operands anywhere in memory, every tile reduces its whole M.  Against torch fp32 on the bf16-rounded operands (as
    test_linear_fwd_bwd), equal to the per-problem launches within fp32 summation order, and bit-identical between two
    launches (no sum crosses workgroups; the bias gradient's in-block reduction runs in row order)."""
    g = torch.Generator().manual_seed(M + N + n)
    Kp, Np = L.cpad(K), L.cpad(N)
    items, refs = [], []
    for i in range(n):
        x = rb(torch.randn(M, K, generator=g))
        dy = rb(torch.randn(M, N, generator=g))
        xp, dyp = torch.zeros(M, Kp), torch.zeros(M, Np)
        xp[:, :K], dyp[:, :N] = x, dy
        pad = torch.empty(17 * (i + 1), device=DEV)          # operands at unrelated addresses
        items.append((xp.to(torch.bfloat16).to(DEV), dyp.to(torch.bfloat16).to(DEV), pad))
        refs.append((dy.t() @ x, dy.sum(0)))
    outs = L.linear_wgrad_group([(a, b) for a, b, _ in items], M, N, K)
    torch.cuda.synchronize()
    first = [(dw.clone(), db.clone()) for dw, db in outs]
    for (dw, db), (rw, rbias), (xc, dyc, _) in zip(outs, refs, items):
        close(dw, rw, name="grouped wgrad")
        close(db, rbias, name="grouped bias grad")
        dw1, db1 = L.linear_wgrad(xc, dyc, M, N, K)
        torch.cuda.synchronize()
        assert (dw - dw1).abs().max().item() <= 1e-3 * rw.abs().max().item() + 1e-6
    for rep in range(3):
        again = L.linear_wgrad_group([(a, b) for a, b, _ in items], M, N, K)
        torch.cuda.synchronize()
        for (dw, db), (dw0, db0) in zip(again, first):
            assert torch.equal(dw, dw0) and torch.equal(db, db0), rep


@pytest.mark.parametrize("flat,ring", [(0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,n", [(7296, 3072, 768, 3), (1100, 768, 3072, 5), (1030, 2304, 768, 2), (1500, 768, 768, 12)])
def test_grouped_linear_wgrad_tile_orders_and_ring_form_agree(M, N, K, n, flat, ring):
    """The grouped launch in its three forms -- round 3's order (one problem per grid.z slice), the flat slab-sharing order
    (default), and the LDS-DMA ring kernel's 128 x 256 tiles (option wgrad_group_ring) -- against torch fp32; each form is
    bitwise reproducible (every tile reduces its whole M)."""
    g = torch.Generator().manual_seed(M + N + n)
    items, refs = [], []
    for i in range(n):
        x = rb(torch.randn(M, K, generator=g))
        dy = rb(torch.randn(M, N, generator=g))
        items.append((x.to(torch.bfloat16).to(DEV), dy.to(torch.bfloat16).to(DEV)))
        refs.append((dy.t() @ x, dy.sum(0)))
    try:
        H.set_option("wgrad_flat", flat)
        H.set_option("wgrad_group_ring", ring)
        outs = [[(dw.clone(), db.clone()) for dw, db in L.linear_wgrad_group(items, M, N, K)] for _ in range(2)]
        torch.cuda.synchronize()
    finally:
        H.set_option("wgrad_flat", 1)
        H.set_option("wgrad_group_ring", H.WGRAD_GROUP_RING_DEFAULT)
    for (dw, db), (dw2, db2), (rw, rbias) in zip(outs[0], outs[1], refs):
        close(dw, rw, name="grouped wgrad")
        close(db, rbias, name="grouped bias grad")
        assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("N", [1, 4, 37, 64, 512])
def test_hardest_negative_loss_opt_in(N):
    """Opt-in extension (BASELINE north_star: "in-batch hardest-negative mining with wavefront-64 argmin"; the REFERENCE has none,
    SURVEY 0.1, and the default stays its all-negatives sum): value and both gradients against the oracle's autograd, fp32
    tolerances; a tie goes to the smaller index; the default TripletLoss is untouched."""
    import pig.loss
    from oracle import model as O
    g = torch.Generator().manual_seed(N)
    V = torch.randn(N, 512, generator=g).requires_grad_()
    A = (0.25 * V.detach() + torch.randn(N, 512, generator=g)).requires_grad_()      # (correlated pairs: some hinges off for large N)
    ref = O.TripletLoss(0.2, hardest=True)(V, A)
    ref.backward()
    Vd, Ad = V.detach().to(DEV).requires_grad_(), A.detach().to(DEV).requires_grad_()
    loss = pig.loss.TripletLoss(0.2, hardest=True)(Vd, Ad)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) <= 1e-6, (loss.item(), ref.item())
    for got, want in ((Vd.grad, V.grad), (Ad.grad, A.grad)):
        assert (got.cpu() - want).abs().max().item() <= 1e-6 + 1e-5 * want.abs().max().item()
    if N >= 4:
        assert loss.item() > 0 and not torch.equal(loss.detach().cpu(), pig.loss.TripletLoss(0.2)(Vd, Ad).detach().cpu())
        assert abs(pig.loss.TripletLoss(0.2)(Vd, Ad).item() - O.TripletLoss(0.2)(V, A).item()) <= 1e-6       # default unchanged
    if N == 4:      # exact tie: audio 1 and 2 identical -> video 0's hardest negative is index 1 (the smaller)
        A2 = A.detach().clone()
        A2[2] = A2[1]
        Vt, At = V.detach().to(DEV).requires_grad_(), A2.to(DEV).requires_grad_()
        pig.loss.TripletLoss(5.0, hardest=True)(Vt, At).backward()
        Vr, Ar = V.detach().clone().requires_grad_(), A2.clone().requires_grad_()
        O.TripletLoss(5.0, hardest=True)(Vr, Ar).backward()          # torch.max returns the first maximal index as well
        assert (Vt.grad.cpu() - Vr.grad).abs().max().item() <= 1e-5 * Vr.grad.abs().max().item() + 1e-6


def test_batched_attention_gemms():
    """Q K^T per (batch, head) with strided operands, softmax, and the per-head transpose."""
    g = torch.Generator().manual_seed(4)
    B, Hn, T, Dh = 2, 12, 50, 64
    D = Hn * Dh
    q = rb(torch.randn(B * T, D, generator=g))
    k = rb(torch.randn(B * T, D, generator=g))
    ref = torch.einsum("bthd,bshd->bhts", q.view(B, T, Hn, Dh), k.view(B, T, Hn, Dh))
    Tp = L.cpad(T)
    S = torch.zeros(B * Hn, T, Tp, dtype=torch.float32, device=DEV)
    H.igemm(q.to(torch.bfloat16).to(DEV), k.to(torch.bfloat16).to(DEV), S, T, T, Dh, H.gather_dense(D), D, Tp,
            nbatch=B * Hn, inner=Hn, a_s=(T * D, Dh), b_s=(T * D, Dh), c_s=(Hn * T * Tp, T * Tp))
    torch.cuda.synchronize()
    close(S.cpu()[:, :, :T].reshape(B, Hn, T, T), ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item(), name="QK^T")
    P = torch.empty(B * Hn, T, Tp, dtype=torch.bfloat16, device=DEV)
    H.softmax_fwd(S, Tp, P, Tp, B * Hn, T, 0.125)
    Sr = (S.cpu()[:, :, :T] * 0.125).requires_grad_()
    Pr = torch.softmax(Sr, dim=-1)
    dP = torch.randn(B * Hn, T, Tp, generator=g)
    Pr.backward(dP[:, :, :T])
    dS = torch.empty_like(P)
    H.softmax_bwd(dP.to(DEV), Tp, P, Tp, dS, B * Hn, T, 0.125)
    torch.cuda.synchronize()
    close(P[:, :, :T], Pr, name="softmax fwd")
    assert (P.float().cpu()[:, :, T:] == 0).all()
    close(dS[:, :, :T], Sr.grad * 0.125, name="softmax bwd")
    Vt = torch.empty(B * Hn, Dh, Tp, dtype=torch.bfloat16, device=DEV)
    kd = k.to(torch.bfloat16).to(DEV)
    H.transpose_bf16(kd, T * D, D, Vt, Hn * Dh * Tp, Tp, B * Hn, T, Dh, inner=Hn, in_s1=Dh, out_s1=Dh * Tp)
    torch.cuda.synchronize()
    ref_t = k.view(B, T, Hn, Dh).permute(0, 2, 3, 1).reshape(B * Hn, Dh, T)
    assert torch.equal(Vt.float().cpu()[:, :, :T], ref_t)
    assert (Vt.float().cpu()[:, :, T:] == 0).all()


class _BNP:
    pass


def test_batchnorm_fwd_bwd():
    g = torch.Generator().manual_seed(2)
    B, C, T, Hh, W = 2, 45, 3, 5, 6
    Cp = L.cpad(C)
    y = rb(torch.randn(B, C, T, Hh, W, generator=g) * 2 + 0.5).requires_grad_()
    res = rb(torch.randn(B, C, T, Hh, W, generator=g))
    bn = torch.nn.BatchNorm3d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    z_ref = F.relu(bn(y) + res)
    dz = rb(torch.randn(z_ref.shape, generator=g))
    z_ref.backward(dz)
    P = _BNP()
    P.weight, P.bias = bn.weight.detach().clone().to(DEV), bn.bias.detach().clone().to(DEV)
    P.running_mean, P.running_var = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yc = to_cl(y.detach(), Cp)
    M = yc.shape[0]
    nblk = 3
    partials = torch.empty(nblk, 2, Cp, device=DEV)
    H.colstats_bf16(yc, M, Cp, partials, nblk)
    z, sv = L.bn_fwd(yc, partials, nblk, M, P, relu=True, residual=to_cl(res, Cp))
    dy, dres, dg, db = L.bn_bwd(to_cl(dz, Cp), yc, z, sv, P.weight, relu=True, want_dres=True)
    torch.cuda.synchronize()
    thw = (T, Hh, W)
    close(from_cl(z, B, thw, C), z_ref, name="bn fwd")
    close(from_cl(dy, B, thw, C), y.grad, name="bn dx")
    close(dg, bn.weight.grad, name="bn dgamma")
    close(db, bn.bias.grad, name="bn dbeta")
    mask = (z_ref.detach() > 0).float()
    close(from_cl(dres, B, thw, C), dz * mask, name="bn dres")
    close(P.running_mean, bn.running_mean, rtol=1e-3, atol=1e-3, name="running mean")
    close(P.running_var, bn.running_var, rtol=1e-3, atol=1e-3, name="running var")
    assert (z.float().cpu()[:, C:] == 0).all()


def test_layernorm_gelu_add():
    g = torch.Generator().manual_seed(3)
    for D in (512, 768):
        rows = 37
        x = rb(torch.randn(rows, D, generator=g) * 1.5 + 0.3).requires_grad_()
        ln = torch.nn.LayerNorm(D)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(D, generator=g) + 0.5)
            ln.bias.copy_(torch.randn(D, generator=g) * 0.1)
        y_ref = ln(x)
        dy = rb(torch.randn(rows, D, generator=g))
        y_ref.backward(dy)
        lnd = torch.nn.LayerNorm(D).to(DEV)
        lnd.load_state_dict(ln.state_dict())
        xd = x.detach().to(torch.bfloat16).to(DEV)
        y, saved = L.layernorm_fwd(xd, lnd)
        dx, dg, db = L.layernorm_bwd(dy.to(torch.bfloat16).to(DEV), xd, lnd, saved)
        torch.cuda.synchronize()
        close(y, y_ref, name="ln fwd")
        close(dx, x.grad, name="ln dx")
        close(dg, ln.weight.grad, name="ln dgamma")
        close(db, ln.bias.grad, name="ln dbeta")
    x = rb(torch.randn(64, 40, generator=g) * 2).requires_grad_()
    y_ref = F.gelu(x)
    dy = rb(torch.randn(64, 40, generator=g))
    y_ref.backward(dy)
    xd = x.detach().to(torch.bfloat16).to(DEV)
    y, dx, s = torch.empty_like(xd), torch.empty_like(xd), torch.empty_like(xd)
    H.gelu_fwd(xd, y)
    H.gelu_bwd(dy.to(torch.bfloat16).to(DEV), xd, dx)
    H.add_bf16(xd, y, s)
    torch.cuda.synchronize()
    close(y, y_ref, name="gelu fwd")
    close(dx, x.grad, name="gelu bwd")
    close(s, x.detach() + rb(y_ref.detach()), name="add")


def test_conv0_groupnorm_gelu():
    g = torch.Generator().manual_seed(5)
    B, Lw = 2, 1205
    T0 = (Lw - 10) // 5 + 1
    wave = 0.1 * torch.randn(B, Lw, generator=g)
    conv = torch.nn.Conv1d(1, 512, 10, 5, bias=False)
    gn = torch.nn.GroupNorm(512, 512)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(512, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(512, generator=g) * 0.1)
    y_ref = F.gelu(gn(conv(wave.unsqueeze(1))))
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    wv, w = wave.to(DEV), conv.weight.detach().reshape(512, 10).to(DEV).contiguous()
    gam, bet = gn.weight.detach().to(DEV), gn.bias.detach().to(DEV)
    stats = torch.zeros(B, 512, 2, device=DEV)
    out = torch.empty(B * T0, 512, dtype=torch.bfloat16, device=DEV)
    H.conv0_stats(wv, B, Lw, T0, w, stats)
    H.conv0_apply(wv, B, Lw, T0, w, stats, gam, bet, 1e-5, out)
    dout = dy.permute(0, 2, 1).reshape(-1, 512).to(torch.bfloat16).to(DEV).contiguous()
    red = torch.zeros(B, 512, 2, device=DEV)
    dw, dgam, dbet = torch.zeros(512, 10, device=DEV), torch.zeros(512, device=DEV), torch.zeros(512, device=DEV)
    H.conv0_bwd_reduce(wv, B, Lw, T0, w, stats, gam, bet, 1e-5, dout, red)
    H.conv0_bwd_apply(wv, B, Lw, T0, w, stats, gam, bet, 1e-5, dout, red, dw, dgam, dbet)
    torch.cuda.synchronize()
    close(out.float().cpu().reshape(B, T0, 512).permute(0, 2, 1), y_ref, name="conv0 fwd")
    close(dw, conv.weight.grad.reshape(512, 10), name="conv0 dw")
    close(dgam, gn.weight.grad, name="conv0 dgamma")
    close(dbet, gn.bias.grad, name="conv0 dbeta")


def test_weightnorm():
    g = torch.Generator().manual_seed(6)
    Co, Ci, K = 32, 8, 16
    v = torch.randn(Co, Ci, K, generator=g).requires_grad_()
    gg = (torch.rand(1, 1, K, generator=g) + 0.5).requires_grad_()
    w_ref = gg * v / v.norm(2, dim=(0, 1), keepdim=True)
    dw = torch.randn(Co, Ci, K, generator=g)
    w_ref.backward(dw)
    vd, gd = v.detach().to(DEV), gg.detach().reshape(K).to(DEV)
    norm = torch.empty(K, device=DEV)
    out = torch.empty(Co, K, Ci, dtype=torch.bfloat16, device=DEV)
    H.weightnorm_fwd(vd, gd, Co, Ci, K, norm, out)
    dwt = dw.permute(0, 2, 1).contiguous().to(DEV)
    dv, dg, ws = torch.empty_like(vd), torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    H.weightnorm_bwd(dwt, vd, gd, norm, Co, Ci, K, dv, dg, ws)
    torch.cuda.synchronize()
    close(out.float().cpu().permute(0, 2, 1), w_ref, name="weightnorm fwd")
    close(dv, v.grad, rtol=1e-4, atol=1e-5, name="weightnorm dv")
    close(dg, gg.grad.reshape(K), rtol=1e-4, atol=1e-5, name="weightnorm dg")


@pytest.mark.parametrize("T,Fd,proj", [(49, 28, True), (2, 512, True), (5, 40, False)])
def test_attnpool_head(T, Fd, proj):
    from oracle import model as O
    g = torch.Generator().manual_seed(T)
    B, Hd = 4, 128
    E = 512 if proj else Fd
    torch.manual_seed(T)
    att = O.Attention(Fd, Hd)
    lin = torch.nn.Linear(Fd, E) if proj else None
    x = torch.randn(B, T, Fd, generator=g).requires_grad_()
    pooled = att(x)
    out_ref = F.normalize(lin(pooled) if proj else pooled, p=2, dim=1)
    dout = torch.randn(B, E, generator=g)
    out_ref.backward(dout)

    def d(t):
        return None if t is None else t.detach().to(DEV).contiguous()

    def e(*s):
        return torch.empty(*s, device=DEV)
    W1, b1, W2, b2 = d(att.hidden.weight), d(att.hidden.bias), d(att.out.weight), d(att.out.bias)
    Wp, bp = (d(lin.weight), d(lin.bias)) if proj else (None, None)
    xd = d(x)
    hid, alpha, pl, pre, out = e(B, T, Hd), e(B, T, Fd), e(B, Fd), e(B, E), e(B, E)
    H.attnpool_fwd(xd, B, T, Fd, Hd, E, W1, b1, W2, b2, Wp, bp, hid, alpha, pl, pre, out)
    dx, dW1, db1, dW2, db2 = e(B, T, Fd), e(Hd, Fd), e(Hd), e(Fd, Hd), e(Fd)
    dWp, dbp = (e(E, Fd), e(E)) if proj else (None, None)
    ws = e(H.attnpool_ws_floats(B, T, Fd, Hd, E))
    H.attnpool_bwd(d(dout), xd, B, T, Fd, Hd, E, W1, W2, Wp, hid, alpha, pl, pre, out, dx, dW1, db1, dW2, db2, dWp,
                   dbp, ws)
    torch.cuda.synchronize()
    tol = dict(rtol=1e-4, atol=1e-5)
    close(out, out_ref, name="head out", **tol)
    close(dx, x.grad, name="head dx", **tol)
    close(dW1, att.hidden.weight.grad, name="dW1", **tol)
    close(db1, att.hidden.bias.grad, name="db1", **tol)
    close(dW2, att.out.weight.grad, name="dW2", **tol)
    close(db2, att.out.bias.grad, name="db2", **tol)
    if proj:
        close(dWp, lin.weight.grad, name="dWp", **tol)
        close(dbp, lin.bias.grad, name="dbp", **tol)


def test_spatial_mean():
    g = torch.Generator().manual_seed(8)
    B, T, HW, C = 2, 3, 49, 512
    x = rb(torch.randn(B, T, HW, C, generator=g))
    xd = x.to(torch.bfloat16).to(DEV)
    out = torch.empty(B, T, C, device=DEV)
    H.spatial_mean_fwd(xd, out, B, T, HW, C, C)
    dout = torch.randn(B, T, C, generator=g)
    dx = torch.empty_like(xd)
    H.spatial_mean_bwd(dout.to(DEV), dx, B, T, HW, C, C)
    torch.cuda.synchronize()
    close(out, x.mean(2), rtol=1e-5, atol=1e-6, name="spatial mean")
    close(dx, (dout / HW).unsqueeze(2).expand(B, T, HW, C), name="spatial mean bwd")


def test_triplet_loss_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "ref_loss.npz"))

    def run(V, A, margin=0.2, dloss=1.0):
        V, A = V.to(DEV).contiguous(), A.to(DEV).contiguous()
        N, D = V.shape
        ws = torch.empty(H.triplet_workspace_bytes(N, D) // 4, device=DEV)
        loss = torch.empty(1, device=DEV)
        H.triplet_loss_fwd(V, A, margin, loss, ws)
        dV, dA = torch.empty_like(V), torch.empty_like(A)
        H.triplet_loss_bwd(V, A, torch.tensor([dloss], device=DEV), ws, dV, dA)
        torch.cuda.synchronize()
        return loss.item(), dV.cpu(), dA.cpu()

    for n in (4, 64):
        loss, dV, dA = run(torch.tensor(d[f"V_{n}"]), torch.tensor(d[f"A_{n}"]))
        assert abs(loss - float(d[f"loss_{n}"])) < 1e-6
        np.testing.assert_allclose(dV.numpy(), d[f"dV_{n}"], atol=2e-7)
        np.testing.assert_allclose(dA.numpy(), d[f"dA_{n}"], atol=2e-7)
    gg = torch.Generator().manual_seed(0)
    V = F.normalize(torch.randn(512, 512, generator=gg))
    A = F.normalize(torch.randn(512, 512, generator=gg))
    loss, dV, dA = run(V, A)
    assert abs(loss - float(d["loss_512"])) < 1e-6
    assert abs(dV.norm().item() - float(d["dVnorm_512"])) < 1e-6
    np.testing.assert_allclose(dV[:4].numpy(), d["dV_512_head"], atol=2e-7)
    np.testing.assert_allclose(dA[-4:].numpy(), d["dA_512_tail"], atol=2e-7)
    loss, dV, dA = run(torch.tensor(d["rawV"]), torch.tensor(d["rawA"]))
    assert abs(loss - float(d["raw_loss"])) < 1e-6
    np.testing.assert_allclose(dV.numpy(), d["raw_dV"], atol=2e-7)
    np.testing.assert_allclose(dA.numpy(), d["raw_dA"], atol=2e-7)
    for m, ref in zip(d["margins"], d["margin_losses"]):
        loss, _, _ = run(torch.tensor(d["V_4"]), torch.tensor(d["A_4"]), margin=float(m))
        assert abs(loss - ref) < 1e-6
    loss, dV2, _ = run(torch.tensor(d["V_64"]), torch.tensor(d["A_64"]), dloss=8.0)
    np.testing.assert_allclose(dV2.numpy(), 8.0 * d["dV_64"], atol=2e-6)


@pytest.mark.parametrize("N", [1, 5, 33, 64, 100, 512])
def test_triplet_loss_ragged_sizes_against_oracle(N):
    """The two-launch loss (prep + 32x32 hinge tiles) at sizes that do not fill its tiles, against the oracle's closed
    form (pinned to the live reference by tests/golden/ref_loss.npz); N = 1 has no off-diagonal term: loss 0, no NaN."""
    from oracle import model as O
    import pig.loss
    g = torch.Generator().manual_seed(N)
    V = (torch.randn(N, 96, generator=g) * 3).requires_grad_()
    A = (torch.randn(N, 96, generator=g) + 0.3 * V.detach()).requires_grad_()
    ref = O.TripletLoss(0.2)(V, A)
    ref.backward()
    Vd, Ad = V.detach().to(DEV).requires_grad_(), A.detach().to(DEV).requires_grad_()
    loss = pig.loss.TripletLoss(0.2)(Vd, Ad)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert (Vd.grad.cpu() - V.grad).abs().max() < 3e-7 and (Ad.grad.cpu() - A.grad).abs().max() < 3e-7


def test_cosine_matrix_and_contrastive_are_differentiable(golden_dir):
    """pig/loss.py:41-55: both are ordinary differentiable torch expressions in the reference.  Values against the live
    reference's golden vectors, gradients against the oracle's autograd, composed == TripletLoss."""
    from oracle import model as O
    import pig.loss
    d = np.load(os.path.join(golden_dir, "ref_loss.npz"))
    for n in (4, 64):
        V, A = torch.tensor(d[f"V_{n}"]), torch.tensor(d[f"A_{n}"])
        Vd, Ad = V.to(DEV).requires_grad_(), A.to(DEV).requires_grad_()
        loss = pig.loss.contrastive(pig.loss.cosine_matrix(Vd, Ad), margin=0.2)
        loss.backward()
        torch.cuda.synchronize()
        assert abs(loss.item() - float(d[f"loss_{n}"])) < 1e-6
        np.testing.assert_allclose(Vd.grad.cpu().numpy(), d[f"dV_{n}"], atol=3e-7)
        np.testing.assert_allclose(Ad.grad.cpu().numpy(), d[f"dA_{n}"], atol=3e-7)
    # rectangular cosine_matrix with an arbitrary upstream gradient; un-normalised rows
    g = torch.Generator().manual_seed(3)
    U, W, R = torch.randn(37, 50, generator=g) * 2, torch.randn(21, 50, generator=g), torch.randn(37, 21, generator=g)
    Ur, Wr = U.clone().requires_grad_(), W.clone().requires_grad_()
    (O.cosine_matrix(Ur, Wr) * R).sum().backward()
    Ud, Wd = U.to(DEV).requires_grad_(), W.to(DEV).requires_grad_()
    S = pig.loss.cosine_matrix(Ud, Wd)
    (S * R.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    close(S.detach(), O.cosine_matrix(U, W), rtol=1e-5, atol=1e-6, name="cosine_matrix")
    close(Ud.grad, Ur.grad, rtol=1e-4, atol=1e-6, name="cosine_matrix dU")
    close(Wd.grad, Wr.grad, rtol=1e-4, atol=1e-6, name="cosine_matrix dV")
    # contrastive on a given matrix, scaled upstream gradient
    M = (torch.randn(19, 19, generator=g) * 0.3)
    Mr = M.clone().requires_grad_()
    (3.0 * O.contrastive(Mr, 0.35)).backward()
    Md = M.to(DEV).requires_grad_()
    out = pig.loss.contrastive(Md, margin=0.35)
    (3.0 * out).backward()
    torch.cuda.synchronize()
    assert abs(out.item() - O.contrastive(M, 0.35).item()) < 1e-6
    close(Md.grad, Mr.grad, rtol=1e-5, atol=1e-7, name="contrastive dM")
    assert not pig.loss.cosine_matrix(Ud.detach(), Wd.detach()).requires_grad


def test_triplet_accuracy_golden_with_exact_tie(golden_dir):
    """pig/metrics.py:45-52 through pp_triplet_accuracy on the live reference's fixture (incl. an exact tie -> 0.5)."""
    import pig.metrics
    d = np.load(os.path.join(golden_dir, "ref_metrics.npz"))
    anc, pos, neg = (torch.tensor(d[k]).to(DEV) for k in ("anchor", "positive", "negative"))
    acc = pig.metrics.triplet_accuracy(anc, pos, neg)
    diff = pig.metrics.triplet_accuracy(anc, pos, neg, discrete=False)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(acc.cpu().numpy(), d["acc"])
    assert acc[5].item() == 0.5
    np.testing.assert_allclose(diff.cpu().numpy(), d["diff"], atol=2e-7)
    assert diff[5].item() == 0.0


def test_bertadam_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "ref_bertadam.npz"))
    ps = [torch.tensor(d[f"p0_{i}"]).to(DEV) for i in range(5)]
    gs = [torch.zeros_like(p) for p in ps]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    tl, keep = H.make_tensor_list(ps, gs, ms, vs, DEV)
    chunk = 16
    ct, co = [], []
    for i, p in enumerate(ps):
        for off in range(0, p.numel(), chunk):
            ct.append(i)
            co.append(off)
    ctd = torch.tensor(ct, dtype=torch.int32, device=DEV)
    cod = torch.tensor(co, dtype=torch.int64, device=DEV)
    norms = torch.empty(5 + len(ct), device=DEV)    # [tensors] + [chunks]: the deterministic mode's per-chunk partials
    from oracle.model import warmup_linear
    for step in range(6):
        for i in range(5):
            gs[i].copy_(torch.tensor(d[f"g{step}_{i}"]))
        lr = 1e-3 * warmup_linear(step / 8, 0.25)
        H.bertadam_step(tl, ctd, cod, len(ct), chunk, norms, lr, 0.9, 0.999, 1e-6, 0.01, 1.0)
        torch.cuda.synchronize()
        for i in range(5):
            np.testing.assert_allclose(ps[i].cpu().numpy(), d[f"p{step + 1}_{i}"], rtol=2e-6, atol=3e-7)  # few fp32 ulps (FMA contraction)
    for i in range(5):
        np.testing.assert_allclose(ms[i].cpu().numpy(), d[f"m_{i}"], atol=1e-6)
        np.testing.assert_allclose(vs[i].cpu().numpy(), d[f"v_{i}"], rtol=5e-5, atol=1e-9)  # fma vs mul+add over 6 steps


def test_maxpool_3x3s2_with_ties():
    g = torch.Generator().manual_seed(12)
    N, C, Hh, W = 3, 16, 9, 12
    x = torch.randint(-3, 4, (N, C, Hh, W), generator=g).float().requires_grad_()   # small integers -> many ties
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    xc = x.detach().permute(0, 2, 3, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous()
    Ho, Wo = y_ref.shape[2:]
    y = torch.empty(N * Ho * Wo, C, dtype=torch.bfloat16, device=DEV)
    H.maxpool3x3s2_fwd(xc, y, N, Hh, W, C)
    dx = torch.empty_like(xc)
    H.maxpool3x3s2_bwd(xc, dy.permute(0, 2, 3, 1).reshape(-1, C).to(torch.bfloat16).to(DEV).contiguous(), dx, N, Hh, W, C)
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu().reshape(N, Ho, Wo, C).permute(0, 3, 1, 2), y_ref.detach())
    close(dx.float().cpu().reshape(N, Hh, W, C).permute(0, 3, 1, 2), x.grad, rtol=1e-2, name="maxpool bwd")


def test_dropout_mask_is_counter_based():
    n, p = 1 << 20, 0.1
    x = torch.ones(n, dtype=torch.bfloat16, device=DEV)
    y1, y2, y3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    H.dropout_bf16(x, y1, p, 1234)
    H.dropout_bf16(x, y2, p, 1234)
    H.dropout_bf16(x, y3, p, 99)
    g = torch.full((n,), 2.0, dtype=torch.float32, device=DEV)
    gf = torch.empty_like(g)
    H.dropout_f32(g, gf, p, 1234)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)           # same seed -> same mask
    keep = (y1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 3e-3
    kept = y1[y1 != 0].float()
    assert (kept - 1 / 0.9).abs().max() < 1e-2                         # scaled by 1/(1-p) (bf16)
    assert torch.equal(gf != 0, y1 != 0)                               # fp32 variant uses the same mask
    r = torch.full((n,), 0.5, dtype=torch.bfloat16, device=DEV)
    yr = torch.empty_like(x)
    H.dropout_bf16(x, yr, p, 1234, res=r)
    torch.cuda.synchronize()
    assert torch.equal((yr.float() - 0.5) != 0, y1 != 0)


def test_batchnorm_bwd_mask_recomputed_from_y():
    """Units without a residual input pass z=None: the ReLU mask comes from y*scale+shift."""
    g = torch.Generator().manual_seed(21)
    B, C, T, Hh, W = 2, 144, 2, 6, 5
    y = rb(torch.randn(B, C, T, Hh, W, generator=g) * 1.5).requires_grad_()
    bn = torch.nn.BatchNorm3d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    z_ref = F.relu(bn(y))
    dz = rb(torch.randn(z_ref.shape, generator=g))
    z_ref.backward(dz)
    P = _BNP()
    P.weight, P.bias = bn.weight.detach().clone().to(DEV), bn.bias.detach().clone().to(DEV)
    P.running_mean, P.running_var = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yc = to_cl(y.detach(), C)
    M = yc.shape[0]
    partials = torch.empty(2, 2, C, device=DEV)
    H.colstats_bf16(yc, M, C, partials, 2)
    z, sv = L.bn_fwd(yc, partials, 2, M, P, relu=True)
    dy1, _, dg1, db1 = L.bn_bwd(to_cl(dz, C), yc, z, sv, P.weight, relu=True)
    dy2, _, dg2, db2 = L.bn_bwd(to_cl(dz, C), yc, None, sv, P.weight, relu=True)
    torch.cuda.synchronize()
    thw = (T, Hh, W)
    close(from_cl(dy2, B, thw, C), y.grad, name="bn dx (mask from y)")
    close(dg2, bn.weight.grad, name="bn dgamma (mask from y)")
    assert (dy1.float() - dy2.float()).abs().max().item() <= 2e-2 * y.grad.abs().max().item()
    assert (db1 - db2).abs().max().item() <= 1e-2 * bn.bias.grad.abs().max().item() + 1e-3


def test_strided_dgrad_with_residual_accumulate():
    """Parity-class decomposition + output row map + residual epilogue (first conv of layer2-4)."""
    g = torch.Generator().manual_seed(33)
    Ci, Co, k, s, p, B, T, Hh, W = 64, 96, (1, 3, 3), (1, 2, 2), (0, 1, 1), 2, 2, 9, 11
    x = rb(torch.randn(B, Ci, T, Hh, W, generator=g)).requires_grad_()
    w = rb(torch.randn(Co, Ci, *k, generator=g) / 24)
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    res = rb(torch.randn(B, Ci, T, Hh, W, generator=g))
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    _, wd = L.prep_conv_weights(w.to(DEV), geom)
    dx = L.conv_dgrad(to_cl(dy, geom.out_cstride), geom, wd, residual=to_cl(res, geom.in_cstride))
    torch.cuda.synchronize()
    close(from_cl(dx, B, (T, Hh, W), Ci), x.grad + res, name="strided dgrad + residual")


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W -- (1,3,3) convolutions with stride (1,2,2): the first convolution of layers 2 / 3 / 4
    (64, 230, 2, 3, 56, 56),      # layer 2.0 at the BASELINE geometry: dy has 240 channels (3.75 chunks of 64), one column block
    (128, 460, 1, 2, 28, 28),     # layer 3.0: two column blocks, 464 channels (7.25 chunks)
    (64, 230, 1, 2, 25, 45),      # odd height AND width (the reference's own 100 x 180 clips at layer 2.0): dx rows / columns without
                                  # a partner in the last dy row / column
    (256, 921, 1, 1, 13, 23),     # layer 4.0 at that geometry: four column blocks, 928 channels (14.5 chunks)
    (64, 96, 2, 2, 9, 11),        # tiny: most of a tile's window lies in other frames
    (64, 230, 1, 1, 40, 126),     # widest supported output (W' + 1 = 64)
])
@pytest.mark.parametrize("with_res", [False, True])
def test_strided_dgrad_window_kernel_matches_parity_classes_and_torch(case, with_res):
    """S2D form of the window kernel (csrc/igemm_win.hip): the stride-(1,2,2) data gradient as ONE launch that reads dy once
    and scatters four output parity classes, against the per-class gather launches it replaces and against torch."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (1, 3, 3), (1, 2, 2), (0, 1, 1)
    g = torch.Generator().manual_seed(Ci + Co + Hh)
    x = rb(torch.randn(B, Ci, T, Hh, W, generator=g)).requires_grad_()
    w = rb(torch.randn(Co, Ci, *k, generator=g) / math.sqrt(9 * Ci))
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    res = rb(torch.randn(B, Ci, T, Hh, W, generator=g))
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    _, wd = L.prep_conv_weights(w.to(DEV), geom)
    dyc = to_cl(dy, geom.out_cstride)
    resc = to_cl(res, geom.in_cstride) if with_res else None
    outs = []
    try:
        for s2d in (False, True):
            L.WIN_S2D = s2d
            prev_thr = H.WIN_IGEMM_DEFAULT
            H.set_option("win_igemm", 1)          # (the tiny cases lie below the default row threshold)
            H.WIN_IGEMM_DEFAULT = 1
            try:
                dx = L.conv_dgrad(dyc, geom, wd, residual=resc)
                torch.cuda.synchronize()
            finally:
                H.WIN_IGEMM_DEFAULT = prev_thr
                H.set_option("win_igemm", prev_thr)
            outs.append(dx.float().cpu())
    finally:
        L.WIN_S2D = True
    ref = x.grad + (res if with_res else 0)
    close(from_cl(outs[1], B, (T, Hh, W), Ci), ref, name="S2D window dgrad vs torch")
    scale = outs[0].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 2.0 ** -7 * scale, "S2D differs from the parity-class launches"
    assert (outs[1][:, Ci:] == 0).all()       # padded channels stay exact zeros


@pytest.mark.parametrize("case", [
    # Ci, Co, k, s, p, B, T, H, W   (sizes that give several 256-row tiles per workgroup and ragged last tiles)
    (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 3, 5, 56, 56),
    (144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 7, 28, 30),
    (128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1, 3, 13, 11),
    (32, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 4, 9, 10),
    (64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), 2, 3, 20, 22),
])
def test_ring_igemm_matches_register_staged(case):
    """The LDS-DMA ring variant (256-row tiles, three-slot ring) is bit-identical to the register-staged kernel:
    same K order, same MFMA sequence per output, same per-128-row statistics partials."""
    Ci, Co, k, s, p, B, T, Hh, W = case
    g = torch.Generator().manual_seed(Ci + 3 * Co)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    x = (torch.randn(geom.Min, geom.in_cstride, generator=g)).to(torch.bfloat16).to(DEV)
    x[:, Ci:] = 0
    dy = (torch.randn(geom.M, geom.out_cstride, generator=g)).to(torch.bfloat16).to(DEV)
    dy[:, Co:] = 0
    w = torch.randn(Co, Ci, *k, generator=g).to(DEV) / math.sqrt(Ci * k[0] * k[1] * k[2])
    wf, wd = L.prep_conv_weights(w, geom)
    lin_w = torch.randn(Co, Ci, generator=g).to(DEV) * 0.05
    lf, lt = L.prep_linear(lin_w)
    xl = torch.randn(geom.M, lf.shape[1], generator=g).to(torch.bfloat16).to(DEV)
    lbias = torch.randn(Co, generator=g).to(DEV)
    res = torch.randn(geom.M, L.cpad(Co), generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    try:
        for ring in (0, 1):
            H.set_option("ring_igemm", ring)
            y, st = L.conv_fwd(x, geom, wf, stats=True)
            dx = L.conv_dgrad(dy, geom, wd)
            yl = L.linear_fwd(xl, geom.M, lf, Co)     # plain-epilogue dense GEMM
            yf = L.linear_fwd(xl, geom.M, lf, Co, bias=lbias, act=H.ACT_GELU, residual=res)   # fused epilogue
            torch.cuda.synchronize()
            # (columns past cpad8(N) are not written)
            outs.append((y.clone(), st.clone(), dx.clone(), yl[:, :Co].clone(), yf[:, :Co].clone()))
    finally:
        H.set_option("ring_igemm", H.RING_IGEMM_DEFAULT)
    for a, b, name in zip(outs[0], outs[1], ("fwd", "colstats", "dgrad", "dense", "dense fused")):
        assert a.shape == b.shape, name
        assert torch.equal(a, b), f"{name}: ring differs from register-staged (max {(a.float() - b.float()).abs().max().item()})"


@pytest.mark.parametrize("shape", [(7296, 768, 768), (1000, 200, 328), (2304, 96, 136)])
def test_dense_ring_tile_widths_agree(shape):
    """Dense ring GEMMs choose 96-, 128- or 144-column tiles by the number of rounds of 256 tiles (pp_set_option
    "ring_wn" forces one): the width changes which workgroup computes an element, not the element -- bit-identical
    outputs, plain and with the fused bias + GELU + residual epilogue; the transformer's M = 7296 x N = 768 shape
    included."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N)
    lf, _ = L.prep_linear((torch.randn(N, K, generator=g) * 0.05).to(DEV))
    x = torch.randn(M, lf.shape[1], generator=g).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, L.cpad(N), generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    try:
        H.set_option("ring_igemm", 1)
        for wn in (8, 6, 9, 0):
            H.set_option("ring_wn", wn)
            y = L.linear_fwd(x, M, lf, N)
            yf = L.linear_fwd(x, M, lf, N, bias=bias, act=H.ACT_GELU, residual=res)
            torch.cuda.synchronize()
            outs.append((y[:, :N].clone(), yf[:, :N].clone()))
    finally:
        H.set_option("ring_wn", 0)
        H.set_option("ring_igemm", H.RING_IGEMM_DEFAULT)
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    close(outs[0][0], x.float()[:, :K] @ lf.float()[:, :K].t(), name="dense ring vs torch")


@pytest.mark.parametrize("case", [
    # Ci, Co, k, s, p, B, T, H, W
    (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 4, 28, 30),     # Kj = 576 = 3 x 192
    (128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1, 3, 13, 11),    # two row blocks of dW, ragged M
    (230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), 2, 6, 9, 9),      # Kj = 720, strided temporal gather
    (48, 160, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 4, 9, 10),      # 27 taps: 256-column tiles
])
def test_ring_wgrad_matches_register_staged(case):
    """LDS-DMA ring weight gradient (192/256-column tiles, inline-asm transposing LDS reads) against the register-staged
    kernel: same products, different fp32 summation order (M split + atomics), so equal to fp32 rounding."""
    Ci, Co, k, s, p, B, T, Hh, W = case
    g = torch.Generator().manual_seed(11 * Ci + Co)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    x = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(DEV)
    x[:, Ci:] = 0
    dy = torch.randn(geom.M, geom.out_cstride, generator=g).to(torch.bfloat16).to(DEV)
    dy[:, Co:] = 0
    xl = torch.randn(geom.M, L.cpad(Ci), generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    try:
        for ring in (0, 1):
            H.set_option("ring_wgrad", ring)
            gw = L.conv_wgrad_raw(x, dy, geom)
            dwl, dbl = L.linear_wgrad(xl, dy, geom.M, Co, xl.shape[1], want_bias=True)
            torch.cuda.synchronize()
            outs.append((gw.clone(), dwl.clone(), dbl.clone()))
    finally:
        H.set_option("ring_wgrad", H.RING_WGRAD_DEFAULT)
    for a, b, name in zip(outs[0], outs[1], ("conv wgrad", "dense wgrad", "bias grad")):
        scale = a.abs().max().item()
        err = (a - b).abs().max().item()
        assert err <= 2e-5 * scale + 1e-6, f"{name}: ring differs by {err} (scale {scale})"
    # and against plain fp32 PyTorch for the dense pair
    ref = dy.float()[:, :Co].t() @ xl.float()
    close(outs[1][1][:, :xl.shape[1]], ref, name="ring dense wgrad vs torch")
    close(outs[1][2], dy.float()[:, :Co].sum(0), name="ring bias grad vs torch")


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W   -- (1,3,3) stride-1 pad-1 convs, the shapes pp_wgrad's sliding-window kernel takes
    (64, 144, 2, 3, 20, 22),      # one channel block, one 144-row block; frame and image borders inside a step
    (128, 288, 2, 2, 14, 13),     # two channel blocks x two row blocks, odd width
    (64, 128, 1, 5, 9, 56),       # 128-row block (8 tiles), the layer-1 width
    (192, 230, 1, 2, 7, 7),       # ragged rows of dW (230 of 240), three channel blocks, tiny image (window >> image)
    (64, 144, 1, 2, 3, 63),       # widest image of the one-group form (W + 1 = one 64-row step)
    (64, 144, 1, 1, 5, 7),        # fewer rows than one step
    (64, 144, 1, 2, 7, 90),       # the reference's own layer-1 width (100 x 180 clips): two 64-row groups on either side
    (128, 288, 1, 1, 3, 127),     # widest supported image (W + 1 = two steps)
    (64, 144, 1, 1, 4, 64),       # first width of the two-group form
])
def test_sliding_window_wgrad_matches_generic(case):
    """Sliding-window weight gradient (X window in LDS, taps = row-shifted views + border masks) against the generic
    gather kernel and against torch's conv3d weight gradient."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    g = torch.Generator().manual_seed(5 * Ci + Co + W)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    xf = rb(torch.randn(B, Ci, T, Hh, W, generator=g))
    dyf = rb(torch.randn(B, Co, T, Hh, W, generator=g))
    x, dy = to_cl(xf, geom.in_cstride), to_cl(dyf, geom.out_cstride)
    outs = []
    try:
        for sw in (0, 1):
            H.set_option("sw_wgrad", sw)
            gw = L.conv_wgrad_raw(x, dy, geom)
            torch.cuda.synchronize()
            outs.append(gw.clone())
    finally:
        H.set_option("sw_wgrad", H.SW_WGRAD_DEFAULT)
    scale = outs[0].abs().max().item()
    err = (outs[0] - outs[1]).abs().max().item()
    assert err <= 2e-5 * scale + 1e-6, f"sliding-window wgrad differs from the generic kernel by {err} (scale {scale})"
    w = torch.zeros(Co, Ci, *k, requires_grad=True)
    F.conv3d(xf, w, stride=s, padding=p).backward(dyf)
    dw = torch.empty(Co, Ci, *k, dtype=torch.float32, device=DEV)
    H.unprep_conv_grad(outs[1], dw, geom.Co, geom.Cig, geom.taps, geom.cg_in)
    close(dw, w.grad, name="sliding-window wgrad vs torch")


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W  -- (1,3,3) stride-1 pad-1 convs: forward Ci -> Co and data gradient Co -> Ci
    (64, 144, 2, 3, 20, 22),      # fwd: one 64-channel chunk, 144 columns; dgrad: three 48-channel chunks, 64 columns
    (128, 288, 1, 2, 28, 28),     # fwd: two chunks, two column blocks; dgrad: six 48-channel chunks, 128 columns
    (64, 230, 1, 3, 9, 56),       # dgrad from 240 padded channels (five 48-channel chunks); layer-1 width
    (256, 576, 1, 1, 14, 14),     # dgrad with 64-channel chunks (576 = 9 x 64); window >> image
    (64, 128, 3, 1, 7, 5),        # tiny images: most of every window lies in neighbouring images; ragged last tile
    (144, 64, 1, 2, 3, 63),       # widest image of the 64-row halo (W + 1 = halo); fwd from 48-channel chunks to 64 columns
    (64, 144, 1, 1, 4, 9),        # a single partial tile (36 rows)
    (64, 144, 1, 2, 7, 90),       # the reference's own layer-1 width (100 x 180 clips): 96-row halo, two-slot weight ring
    (144, 64, 1, 1, 5, 95),       # widest supported image (W + 1 = 96); fwd 48-channel chunks -> 64 columns, dgrad 64 -> 144
    (128, 128, 1, 1, 4, 64),      # first width of the 96-row halo; 128-column tiles both ways
    (64, 45, 1, 1, 6, 70),        # narrow output from 64-channel chunks; dgrad from 48 padded channels
    (256, 460, 1, 2, 14, 14),     # layer 3's mid-planes: the dgrad reduces over 464 channels = 7.25 chunks of 64 (partial last chunk)
    (512, 921, 1, 1, 7, 7),       # layer 4's: 928 channels = 14.5 chunks
    (208, 96, 1, 1, 9, 12),       # a forward from 208 channels (3.25 chunks) as well
])
def test_window_igemm_matches_gather_igemm(case):
    """Window conv kernel (A halo window in LDS, taps = address offsets, zero row outside the image) against the
    gather kernel: forward with BatchNorm statistics, data gradient with and without the fused residual add."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    g = torch.Generator().manual_seed(3 * Ci + Co + Hh)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    x = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(DEV)
    x[:, Ci:] = 0
    dy = torch.randn(geom.M, geom.out_cstride, generator=g).to(torch.bfloat16).to(DEV)
    dy[:, Co:] = 0
    res = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn(Co, Ci, *k, generator=g).to(DEV) / math.sqrt(Ci * 9)
    wf, wd = L.prep_conv_weights(w, geom)
    outs = []
    try:
        # gather kernel; window kernel with 256-row tiles; window kernel with the 512-row tile forced where it exists
        for win, tall in ((0, 0), (1, 0), (1, 2)):
            H.set_option("win_igemm", win)
            H.set_option("win_tall", tall)
            y, st = L.conv_fwd(x, geom, wf, stats=True)
            dx = L.conv_dgrad(dy, geom, wd)
            dxr = L.conv_dgrad(dy, geom, wd, residual=res)
            torch.cuda.synchronize()
            outs.append((y.float(), st.clone(), dx.float(), dxr.float()))
    finally:
        H.set_option("win_igemm", H.WIN_IGEMM_DEFAULT)
        H.set_option("win_tall", H.WIN_TALL_DEFAULT)
    for a, b, name in list(zip(outs[0], outs[1], ("fwd", "colstats", "dgrad", "dgrad + residual"))) + \
            list(zip(outs[0], outs[2], ("fwd (tall)", "colstats", "dgrad (tall)", "dgrad + residual (tall)"))):
        assert a.shape == b.shape, name
        scale = a.abs().max().item()
        err = (a - b).abs().max().item()
        # same products, fp32 sums in a different order: at most one bf16 rounding step apart (stats: fp32 sums)
        tol = (2e-4 if name == "colstats" else 2.0 ** -7) * scale
        assert err <= tol, f"{name}: window kernel differs by {err} (scale {scale})"
        assert (a - b).abs().mean().item() <= 1e-3 * scale, name
    # ... and directly against torch's fp32 conv3d on the same 16-bit-rounded operands (the comparison
    # above alone would pass if both kernels shared a defect)
    xf = from_cl(x, B, (T, Hh, W), Ci).requires_grad_()
    wr = rb(w.cpu()).requires_grad_()
    yr = F.conv3d(xf, wr, stride=s, padding=p)
    dyf = from_cl(dy, B, (T, Hh, W), Co)
    yr.backward(dyf)
    for idx, tag in ((1, "256-row tiles"), (2, "tall")):
        close(from_cl(outs[idx][0], B, (T, Hh, W), Co), yr.detach(), name=f"window fwd vs torch ({tag})")
        close(from_cl(outs[idx][2], B, (T, Hh, W), Ci), xf.grad, name=f"window dgrad vs torch ({tag})")
        close(from_cl(outs[idx][3], B, (T, Hh, W), Ci), xf.grad + from_cl(res, B, (T, Hh, W), Ci), name=f"window dgrad + residual vs torch ({tag})")


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W  -- (3,1,1) stride-1 pad-(1,0,0) convs with 4 / 8 / 16 / 32 frames: tiles = all frames x positions
    (144, 64, 2, 16, 8, 8),       # layer-1 temporal: fwd from three 48-channel chunks, dgrad 64 -> 144 columns
    (45, 64, 3, 16, 4, 12),       # stem: 45 (48) input channels; dgrad to 45 of 48 columns; three clips
    (144, 64, 1, 8, 8, 12),       # 8 frames: 32 positions per tile
    (64, 64, 2, 4, 8, 8),         # 4 frames: 64 positions per tile; 64-channel chunks both ways
    (144, 64, 1, 16, 56, 56),     # layer-1 frame size (196 tiles per clip)
    (144, 64, 2, 32, 8, 12),      # 32 frames (BASELINE configs[4] clips): 8 positions per tile
    (45, 64, 1, 32, 4, 14),       # stem at 32 frames
    (144, 64, 2, 23, 10, 18),     # the reference's own 23 frames (round 4): 11 positions per tile, 3 dead tile rows, ragged last block
    (45, 64, 1, 23, 50, 90),      # ... at its layer-1 frame size, stem channels
    (64, 64, 3, 12, 5, 9),        # 12 frames (layer 2 of those clips): 21 positions per tile, 45 positions per frame
    (144, 64, 4, 6, 7, 7),        # 6 frames, 42 positions per tile, 49 per frame: a full and a ragged tile per clip
    (64, 144, 5, 3, 7, 12),       # 3 frames (layer 4): 85 positions per tile; 144 output columns (no statistics in that form)
    (144, 64, 3, 31, 4, 4),       # 31 frames: 8 positions per tile, 8 dead rows
    (64, 64, 8, 2, 9, 9),         # 2 frames: taps -1 and T both masked for every row
])
def test_temporal_window_igemm_matches_gather_igemm(case):
    """Window kernel in its temporal form (tile = every frame of a block of positions, taps = +-block rows, frames -1
    and T masked to the zero row) against the gather kernel: forward with BatchNorm statistics, data gradient with
    and without the fused residual add, and both against torch."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (3, 1, 1), (1, 1, 1), (1, 0, 0)
    g = torch.Generator().manual_seed(5 * Ci + Co + T)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    xf = torch.randn(B, Ci, T, Hh, W, generator=g)
    dyf = torch.randn(B, Co, T, Hh, W, generator=g)
    x = to_cl(xf, geom.in_cstride)
    dy = to_cl(dyf, geom.out_cstride)
    res = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * 3)
    wf, wd = L.prep_conv_weights(w.to(DEV), geom)
    outs = []
    try:
        for win in (0, 1):
            H.set_option("win_temporal", win)
            y, st = L.conv_fwd(x, geom, wf, stats=True)
            dx = L.conv_dgrad(dy, geom, wd)
            dxr = L.conv_dgrad(dy, geom, wd, residual=res)
            torch.cuda.synchronize()
            outs.append((y.float(), st.sum(dim=0), dx.float(), dxr.float()))      # (partials cover different row sets)
    finally:
        H.set_option("win_temporal", 1)
    for a, b, name in zip(outs[0], outs[1], ("fwd", "colstats", "dgrad", "dgrad + residual")):
        assert a.shape == b.shape, name
        scale = a.abs().max().item()
        err = (a - b).abs().max().item()
        tol = (2e-4 if name == "colstats" else 2.0 ** -7) * scale
        assert err <= tol, f"{name}: temporal window kernel differs by {err} (scale {scale})"
        assert (a - b).abs().mean().item() <= 1e-3 * scale, name
    xr = xf.to(torch.bfloat16).float().requires_grad_()
    wr = w.to(torch.bfloat16).float()
    yr = F.conv3d(xr, wr, stride=s, padding=p)
    yr.backward(dyf.to(torch.bfloat16).float())
    close(from_cl(outs[1][0].to(torch.bfloat16), B, (T, Hh, W), Co), yr.detach(), name="temporal window fwd vs torch")
    close(from_cl(outs[1][2].to(torch.bfloat16), B, (T, Hh, W), Ci), xr.grad, name="temporal window dgrad vs torch")


@pytest.mark.parametrize("case", [
    # Ci, Co, k, p, B, T, H, W, with_residual, with_z     (data gradient Co -> Ci; the consumer's BatchNorm has Ci channels)
    (64, 144, (1, 3, 3), (0, 1, 1), 2, 3, 20, 22, False, False),    # spatial window, 64 columns, 512-row tiles when forced
    (64, 144, (1, 3, 3), (0, 1, 1), 2, 3, 20, 22, True, True),      # ... with the skip gradient added and the mask taken from z
    (128, 288, (1, 3, 3), (0, 1, 1), 1, 2, 28, 28, False, False),   # 128 columns (WN = 8), 48-channel chunks
    (256, 576, (1, 3, 3), (0, 1, 1), 4, 2, 14, 14, True, False),    # 256 columns: two column blocks per row tile
    (64, 64, (3, 1, 1), (1, 0, 0), 2, 8, 8, 16, False, False),      # temporal window, 64 columns
    (45, 64, (3, 1, 1), (1, 0, 0), 2, 16, 4, 12, False, False),     # temporal window, 45 of 48 columns
    (64, 230, (1, 3, 3), (0, 1, 1), 1, 3, 9, 56, True, True),       # ragged last tile
])
@pytest.mark.parametrize("tall", [0, 2])
def test_bn_backward_sums_in_the_dgrad_epilogue(case, tall):
    """pp_igemm_desc.bnr_*: the window data-gradient kernels accumulate sum g and sum g * xhat of the BatchNorm layer that
    receives their output as dz; the sums must equal those pp_bn_bwd_reduce takes from the stored dz."""
    Ci, Co, k, p, B, T, Hh, W, with_res, with_z = case
    g = torch.Generator().manual_seed(Ci + 3 * Co + T)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, (1, 1, 1), p)
    dy = to_cl(rb(torch.randn(B, Co, T, Hh, W, generator=g)), geom.out_cstride)
    w = rb(torch.randn(Co, Ci, *k, generator=g) / math.sqrt(Ci * k[0] * k[1] * k[2]))
    _, wd = L.prep_conv_weights(w.to(DEV).contiguous(), geom)
    Cp = geom.in_cstride
    y = to_cl(rb(torch.randn(B, Ci, T, Hh, W, generator=g)), Cp)
    z = to_cl(rb(torch.randn(B, Ci, T, Hh, W, generator=g)), Cp) if with_z else None
    res = to_cl(rb(torch.randn(B, Ci, T, Hh, W, generator=g)), Cp) if with_res else None
    sv = L.BNSaved()
    sv.count, sv.C, sv.Cp = geom.Min, Ci, Cp
    pad = lambda t: torch.cat([t, torch.zeros(Cp - Ci)]).to(DEV)
    sv.mean, sv.rstd = pad(0.3 * torch.randn(Ci, generator=g)), pad(0.5 + torch.rand(Ci, generator=g))
    sv.scale, sv.shift = pad(torch.randn(Ci, generator=g)), pad(0.3 * torch.randn(Ci, generator=g))
    try:
        H.set_option("win_tall", tall)
        L.FUSE_BN_BWD_REDUCE = True      # (off by default: the fused epilogue measured 0.6 ms per step slower)
        for relu in (True, False):
            dx = L.conv_dgrad(dy, geom, wd, residual=res, consumer=(y, z, sv, relu))
            torch.cuda.synchronize()
            assert hasattr(dx, "_bnr"), "this shape should have taken the fused path"
            partials, nblk = dx._bnr
            assert nblk == (geom.Min + 255) // 256 and partials.shape == (nblk, 2, Cp)
            plain = L.conv_dgrad(dy, geom, wd, residual=res)
            assert torch.equal(plain, dx), "the product itself must not change"
            nref = min(2048, (geom.Min + 63) // 64)
            ref = torch.empty(nref, 2, Cp, device=DEV)
            H.bn_bwd_reduce(dx, y, z, sv.mean, sv.rstd, sv.scale, sv.shift, relu, ref, nref, geom.Min, Cp)
            torch.cuda.synchronize()
            got, want = partials.double().sum(0).cpu(), ref.double().sum(0).cpu()
            scale = want.abs().max().item()
            assert (got - want).abs().max().item() <= 2e-5 * scale + 1e-4, (relu, (got - want).abs().max().item(), scale)
            assert (got[:, Ci:] == 0).all()
        # a kernel without the epilogue says so and the caller falls back (strided / grouped / tiny problems)
        small = L.ConvGeom(32, (2, 6, 6), 32, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        _, wds = L.prep_conv_weights(torch.randn(48, 32, 3, 3, 3, device=DEV), small)
        ys = to_cl(torch.randn(32, 32, 2, 6, 6), small.in_cstride)
        svs = L.BNSaved()
        svs.mean = svs.rstd = svs.scale = svs.shift = torch.ones(small.in_cstride, device=DEV)
        dxs = L.conv_dgrad(to_cl(torch.randn(32, 48, 2, 6, 6), small.out_cstride), small, wds, consumer=(ys, None, svs, True))
        assert not hasattr(dxs, "_bnr")
    finally:
        H.set_option("win_tall", H.WIN_TALL_DEFAULT)
        L.FUSE_BN_BWD_REDUCE = False


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W  -- (3,1,1) stride-1 pad-(1,0,0) convs, the shapes pp_wgrad's temporal window kernel takes
    (144, 64, 2, 5, 8, 8),        # one 144-channel block, 64 rows of dW; frames of exactly 64 positions
    (288, 128, 1, 4, 9, 10),      # two channel blocks, ragged 64-position blocks (90 positions per frame)
    (230, 128, 2, 3, 5, 5),       # 240 padded channels: second channel block partly empty; frames smaller than a block
    (576, 256, 1, 2, 7, 7),       # two row blocks of dW, T = 2 (every step has a tap outside the clip)
    (45, 64, 2, 9, 9, 10),        # the stem's width: the narrow form (48-channel blocks, six steps of look-ahead), ragged blocks
    (48, 64, 1, 3, 8, 8),         # narrow form, a clip shorter than the look-ahead
])
def test_temporal_window_wgrad_matches_generic(case):
    """Temporal sliding-window weight gradient (X blocks of frames t-1, t, t+1 kept in an LDS ring while a 64-position
    column is walked through time) against the generic gather kernel and torch's conv3d weight gradient."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (3, 1, 1), (1, 1, 1), (1, 0, 0)
    g = torch.Generator().manual_seed(7 * Ci + Co + T)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    xf = rb(torch.randn(B, Ci, T, Hh, W, generator=g))
    dyf = rb(torch.randn(B, Co, T, Hh, W, generator=g))
    x, dy = to_cl(xf, geom.in_cstride), to_cl(dyf, geom.out_cstride)
    outs = []
    try:
        for sw in (0, 1):
            H.set_option("sw_wgrad", sw)
            gw = L.conv_wgrad_raw(x, dy, geom)
            torch.cuda.synchronize()
            outs.append(gw.clone())
    finally:
        H.set_option("sw_wgrad", H.SW_WGRAD_DEFAULT)
    scale = outs[0].abs().max().item()
    err = (outs[0] - outs[1]).abs().max().item()
    assert err <= 2e-5 * scale + 1e-6, f"temporal window wgrad differs from the generic kernel by {err} (scale {scale})"
    w = torch.zeros(Co, Ci, *k, requires_grad=True)
    F.conv3d(xf, w, stride=s, padding=p).backward(dyf)
    dw = torch.empty(Co, Ci, *k, dtype=torch.float32, device=DEV)
    H.unprep_conv_grad(outs[1], dw, geom.Co, geom.Cig, geom.taps, geom.cg_in)
    close(dw, w.grad, name="temporal window wgrad vs torch")


def _ref_recall_at_1_to_n(candidates, references, correct, N):
    """The reference's algorithm (pig/metrics.py:23-42) restated with plain torch on the CPU."""
    cn = candidates / candidates.norm(dim=1, keepdim=True)
    rn = references / references.norm(dim=1, keepdim=True)
    distances = 1 - rn @ cn.t()
    recall = [[0.0 for _ in distances]] + [[] for _ in range(N)]
    for j, row in enumerate(distances):
        ranked = row.argsort(stable=True)
        target = torch.nonzero(correct[j])[:, 0]
        for n in range(1, N + 1):
            overlap = (ranked[:n].unsqueeze(0) == target.unsqueeze(1)).sum().item()
            recall[n].append(overlap / len(target))
    return torch.tensor(recall)


def test_recall_metrics_on_device(golden_dir):
    """pig.metrics recall_at_n / recall_at_1_to_n / resampled_recall on the HIP path: the reference's golden vectors
    (generated by the live reference, oracle/make_golden.py), multi-target rows, and the batched resampled variant."""
    import pig.metrics as M
    z = np.load(os.path.join(golden_dir, "ref_metrics.npz"))
    cand, ref = torch.from_numpy(z["cand"]).to(DEV), torch.from_numpy(z["ref"]).to(DEV)
    r3 = M.recall_at_n(cand, ref, torch.eye(16), n=3)
    r14 = M.recall_at_1_to_n(cand, ref, torch.eye(16), N=4)
    assert torch.equal(r3, torch.from_numpy(z["recall_at_3"]).float())
    assert torch.equal(r14, torch.from_numpy(z["recall_1_to_4"]).float())
    # several targets per row, rectangular problem
    g = torch.Generator().manual_seed(9)
    C, R = torch.randn(37, 64, generator=g), torch.randn(23, 64, generator=g)
    correct = (torch.rand(23, 37, generator=g) < 0.15)
    correct[:, 0] = True   # every row has at least one target
    got = M.recall_at_1_to_n(C.to(DEV), R.to(DEV), correct.float(), N=6)
    assert torch.allclose(got, _ref_recall_at_1_to_n(C, R, correct, 6), atol=1e-6)
    # resampled: same index draws as the reference (global CPU generator), one launch for all sets
    X, Y = torch.randn(150, 32, generator=g), torch.randn(150, 32, generator=g)
    Y = Y + 0.5 * X   # make the diagonal retrievable
    torch.manual_seed(123)
    got = M.resampled_recall(X.to(DEV), Y.to(DEV), size=100, n_samples=7, n=5)
    torch.manual_seed(123)
    want = []
    for _ in range(7):
        ix = torch.randperm(150)[:100]
        want.append(_ref_recall_at_1_to_n(X[ix], Y[ix], torch.eye(100), 5)[5])
    assert got.shape == (7, 100) and torch.allclose(got, torch.stack(want), atol=1e-6)
    torch.manual_seed(5)
    t = M.resampled_recall_at_1_to_n(X.to(DEV), Y.to(DEV), size=100, n_samples=3, N=4)
    assert t.shape == (3, 5, 100) and (t[:, 0] == 0).all() and (t[:, 1:].diff(dim=1) >= 0).all()


@pytest.mark.parametrize("B,T,p", [(3, 114, 0.0), (2, 49, 0.1), (2, 128, 0.1), (1, 7, 0.0),
                                   (2, 229, 0.0), (2, 229, 0.1), (1, 256, 0.1), (1, 129, 0.0), (2, 200, 0.0),
                                   (2, 316, 0.0), (1, 316, 0.1), (1, 257, 0.0), (1, 320, 0.1)])
def test_fused_attention_matches_unfused_and_torch(B, T, p):
    """pp_attention_fwd / _bwd (scores in MFMA accumulators, probabilities recomputed in the backward from the forward's
    log-sum-exp rows; T <= 128: one workgroup per clip and head, T <= 320 -- the 229 frames of BASELINE configs[4], the 316
    frames of the reference's own 2.3-s clips at 44.1 kHz -- 64-query-row workgroups forward and 2 x 2 / 3 x 3 blocks of 128
    backward) against the unfused HIP path (same dropout stream) and, without dropout, against fp32 torch."""
    from peppa_amd import audio as A
    g = torch.Generator().manual_seed(B * 100 + T)
    M, Tp = B * T, L.cpad(T)
    qkv = (0.5 * torch.randn(M, 2304, generator=g)).to(torch.bfloat16).to(DEV)
    dctx = torch.randn(M, 768, generator=g).to(torch.bfloat16).to(DEV)
    scale, drop = 0.125, (p, 12345)
    outs = []
    try:
        for fused in (False, True):
            A.FUSED_ATTENTION = fused
            ctx, P = A._attention_fwd(qkv, B, T, Tp, scale, True, drop)
            dqkv = A._attention_bwd(dctx, qkv, ctx, P, B, T, Tp, scale, drop)
            torch.cuda.synchronize()
            assert (P.dtype == torch.float32 and P.shape == (B * 12, T)) == fused   # fused: the log-sum-exp rows
            outs.append((ctx.float().cpu(), dqkv.float().cpu()))
    finally:
        A.FUSED_ATTENTION = True
    for a, b, name in zip(outs[0], outs[1], ("ctx", "dqkv")):
        scale_ = a.abs().max().item()
        assert (a - b).abs().max().item() <= 2.0 ** -6 * scale_, f"{name}: fused differs from unfused by {(a - b).abs().max().item()}"
        assert rel_l2(b, a) <= 6e-3, f"{name}: {rel_l2(b, a)}"
    if p == 0.0:
        x = qkv.float().cpu().view(B, T, 3, 12, 64).requires_grad_()
        q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)   # (B, 12, T, 64)
        ctx_ref = (torch.softmax(scale * q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(M, 768)
        ctx_ref.backward(dctx.float().cpu())
        close(outs[1][0], ctx_ref, name="fused ctx vs torch")
        close(outs[1][1], x.grad.reshape(M, 2304), name="fused dqkv vs torch")


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("T", [321, 600, 1000])
def test_unfused_attention_beyond_the_fused_kernels_reach(T):
    """Clips longer than the fused kernels' 320 frames take batched GEMMs + pp_softmax_* (T <= 1024) on their own: against
    fp32 torch.  (Round 3 refused T > 256 outright -- the reference's own clips have 316 frames.)"""
    from peppa_amd import audio as A
    B = 1
    g = torch.Generator().manual_seed(T)
    M, Tp = B * T, L.cpad(T)
    qkv = (0.5 * torch.randn(M, 2304, generator=g)).to(torch.bfloat16).to(DEV)
    dctx = torch.randn(M, 768, generator=g).to(torch.bfloat16).to(DEV)
    ctx, P = A._attention_fwd(qkv, B, T, Tp, 0.125, True, (0.0, 0))
    assert P.dtype != torch.float32          # the saved probabilities: the unfused path
    dqkv = A._attention_bwd(dctx, qkv, ctx, P, B, T, Tp, 0.125, (0.0, 0))
    torch.cuda.synchronize()
    x = qkv.float().cpu().view(B, T, 3, 12, 64).requires_grad_()
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    ctx_ref = (torch.softmax(0.125 * q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(M, 768)
    ctx_ref.backward(dctx.float().cpu())
    close(ctx.float().cpu(), ctx_ref, name="unfused ctx vs torch")
    close(dqkv.float().cpu(), x.grad.reshape(M, 2304), name="unfused dqkv vs torch")


def test_gelu_bwd_with_fused_dropout_matches_two_passes():
    """dx = dropout_bwd(dy) * gelu'(x) in one pass uses the mask of pp_dropout_bf16 for the same (p, seed)."""
    g = torch.Generator().manual_seed(5)
    M, N = 333, 3072
    dy = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
    x = (2 * torch.randn(M, N, generator=g)).to(torch.bfloat16).to(DEV)
    for p_, seed in ((0.1, 1234567), (0.5, 0xFFFFFFFF), (0.0, 7)):
        masked = torch.empty_like(dy)
        H.dropout_bf16(dy, masked, p_, seed)
        two = torch.empty_like(dy)
        H.gelu_bwd(masked, x, two)
        one = torch.empty_like(dy)
        H.gelu_bwd_dropout(dy, x, one, p_, seed)
        torch.cuda.synchronize()
        assert torch.equal(one == 0, two == 0) or ((one == 0) ^ (two == 0)).float().mean().item() < 1e-3   # same mask
        kept = (masked != 0)
        assert abs(kept.float().mean().item() - (1 - p_)) < 0.01
        # the two-pass route rounds the masked gradient to bf16 before the GELU factor: one rounding step apart
        err = (one.float() - two.float()).abs().max().item()
        assert err <= 2.0 ** -7 * two.float().abs().max().item() + 1e-6, err


@pytest.mark.parametrize("M,N,K,act,with_res", [(1274, 768, 768, "none", True), (7296, 3072, 768, "gelu", False),
                                                 (300, 768, 3072, "none", True), (77, 96, 64, "none", False)])
def test_gemm_epilogue_dropout_matches_separate_dropout(M, N, K, act, with_res):
    """C = dropout(act(x W^T + b)) + residual in the GEMM epilogue carries the very mask pp_dropout_bf16 applies to the flat
    output for the same (p, seed) -- the backward pass regenerates it from (seed, element index)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, L.cpad(N), generator=g).to(torch.bfloat16).to(DEV) if with_res else None
    wf, _ = L.prep_linear(w)
    a = H.ACT_GELU if act == "gelu" else H.ACT_NONE
    p_, seed = 0.1, 0xC0FFEE
    plain = L.linear_fwd(x, M, wf, N, bias=b, act=a)
    two = torch.empty_like(plain)
    H.dropout_bf16(plain, two, p_, seed, res=res)
    pre = torch.empty_like(plain) if act == "gelu" else None
    one = L.linear_fwd(x, M, wf, N, bias=b, act=a, residual=res, pre=pre, dropout=(p_, seed))
    torch.cuda.synchronize()
    mask_ref = torch.empty_like(plain)
    H.dropout_bf16(plain, mask_ref, p_, seed)                 # (without the residual: zero <=> dropped)
    torch.cuda.synchronize()
    one, two, plain = one[:, :N].float(), two[:, :N].float(), plain[:, :N].float()
    r = res[:, :N].float() if with_res else torch.zeros_like(one)
    dropped = (mask_ref[:, :N] == 0) & (plain != 0)
    assert abs(dropped.float().mean().item() - p_) < 0.02
    assert torch.equal((one - r)[dropped], torch.zeros_like(one)[dropped])          # same elements are zeroed
    scale = two.abs().max().item()
    assert (one - two).abs().max().item() <= 2.0 ** -6 * scale          # the two-pass route rounds once more
    if pre is not None:       # the pre-activation copy is never masked
        nodrop_pre = torch.empty_like(pre)
        L.linear_fwd(x, M, wf, N, bias=b, act=a, pre=nodrop_pre)
        torch.cuda.synchronize()
        assert torch.equal(pre, nodrop_pre)


def test_sync_bn_two_ranks_in_one_process_match_the_global_batch():
    """SyncBN (SURVEY 8e option): with the (sum, sum of squares) rows of the forward pass and the (sum g, sum g*xhat) rows of
    the backward pass summed over ranks, every rank reproduces BatchNorm over the global batch.  The all-reduce is played
    by a stand-in that adds the other half's recorded row."""
    class BN:
        pass
    g = torch.Generator().manual_seed(17)
    M, C = 2 * 1536, 144
    Cp = L.cpad(C)
    y = torch.zeros(M, Cp)
    y[:, :C] = torch.randn(M, C, generator=g) * torch.linspace(0.5, 2.0, C) + torch.linspace(-1, 1, C)
    y[: M // 2] += 0.7                       # the halves have different statistics
    dz = torch.zeros(M, Cp)
    dz[:, :C] = torch.randn(M, C, generator=g)
    y, dz = y.to(torch.bfloat16).to(DEV), dz.to(torch.bfloat16).to(DEV)

    def make_bn():
        bn = BN()
        bn.weight = (1 + 0.1 * torch.arange(C, dtype=torch.float32)).to(DEV) / 8
        bn.bias = torch.linspace(-0.5, 0.5, C).to(DEV)
        bn.running_mean, bn.running_var = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        return bn

    def stats(t):
        nb = (t.shape[0] + 127) // 128
        part = torch.zeros(nb, 2, Cp, device=DEV)
        H.colstats_bf16(t, t.shape[0], Cp, part, nb)
        return part, nb

    # reference: one rank holding the whole batch
    bn = make_bn()
    part, nb = stats(y)
    z_ref, sv = L.bn_fwd(y, part, nb, M, bn, relu=True)
    dy_ref, _, dg_ref, db_ref = L.bn_bwd(dz, y, None, sv, bn.weight, relu=True)
    torch.cuda.synchronize()

    halves = [(y[: M // 2].contiguous(), dz[: M // 2].contiguous()), (y[M // 2:].contiguous(), dz[M // 2:].contiguous())]
    rows = {}

    def recorder(key):
        def f(t):
            rows[key] = t.clone()
        return f

    def adder(key):
        def f(t):
            t.add_(rows[key])
        return f
    try:
        L.SYNC_BN_WORLD = 2
        bns = [make_bn(), make_bn()]
        for r, (yh, _) in enumerate(halves):                       # record the forward rows (local sums)
            L.SYNC_BN_REDUCE = recorder(("f", r))
            L.bn_fwd(yh, *stats(yh), M // 2, make_bn(), relu=True)
        fwd = []
        for r, (yh, _) in enumerate(halves):                       # forward with the other rank's row added
            L.SYNC_BN_REDUCE = adder(("f", 1 - r))
            fwd.append(L.bn_fwd(yh, *stats(yh), M // 2, bns[r], relu=True))
        for r, (yh, dh) in enumerate(halves):                      # record the backward rows (they use the global mean / rstd)
            L.SYNC_BN_REDUCE = recorder(("b", r))
            L.bn_bwd(dh, yh, None, fwd[r][1], bns[r].weight, relu=True)
        bwd = []
        for r, (yh, dh) in enumerate(halves):
            L.SYNC_BN_REDUCE = adder(("b", 1 - r))
            bwd.append(L.bn_bwd(dh, yh, None, fwd[r][1], bns[r].weight, relu=True))
        torch.cuda.synchronize()
    finally:
        L.SYNC_BN_REDUCE, L.SYNC_BN_WORLD = None, 1
    z = torch.cat([fwd[0][0], fwd[1][0]])
    dy = torch.cat([bwd[0][0], bwd[1][0]])
    assert fwd[0][1].count == M and torch.allclose(fwd[0][1].mean, sv.mean, atol=1e-5) and torch.allclose(fwd[1][1].rstd, sv.rstd, rtol=1e-5)
    assert (z.float() - z_ref.float()).abs().max().item() <= 2.0 ** -7 * z_ref.float().abs().max().item()
    assert (dy.float() - dy_ref.float()).abs().max().item() <= 2.0 ** -7 * dy_ref.float().abs().max().item()
    assert torch.allclose(bwd[0][2] + bwd[1][2], dg_ref, rtol=1e-4, atol=1e-3)      # per-rank sums add up to the global ones
    assert torch.allclose(bwd[0][3] + bwd[1][3], db_ref, rtol=1e-4, atol=1e-3)
    for r in range(2):                                             # every rank tracks the global running statistics
        assert torch.allclose(bns[r].running_mean, bn.running_mean, atol=1e-5)
        assert torch.allclose(bns[r].running_var, bn.running_var, rtol=1e-4)
    # and the statistics really differ from per-rank ones
    local = L.bn_fwd(halves[0][0], *stats(halves[0][0]), M // 2, make_bn(), relu=True)[1]
    assert (local.mean - sv.mean).abs().max().item() > 0.1


@pytest.mark.parametrize("case", [
    # Ci, Co, B, T, H, W  -- temporal (3,1,1) convs fed by a BatchNorm unit whose apply pass the kernels do themselves
    (144, 64, 2, 16, 8, 8),       # layer-1 width: three 48-channel chunks
    (144, 64, 1, 8, 8, 16),       # 8 frames
    (144, 64, 1, 16, 56, 56),     # layer-1 frame size
    (45, 64, 2, 16, 8, 16),       # the stem: 45 channels in one 48-channel chunk (pad channels: scale = shift = 0)
    (144, 64, 2, 23, 10, 18),     # the reference's own 23 frames: ragged tiles (dead rows are activated too, and must stay out of
                                  # the statistics)
    (45, 64, 1, 23, 12, 20),      # ... the stem's channels
    (144, 64, 1, 12, 9, 25),      # 12 frames: 21 positions per tile, 225 per frame
])
def test_fused_batchnorm_apply_matches_the_separate_pass(case):
    """pp_igemm(a_bn_*) / pp_wgrad(x_bn_*): the temporal window kernel and the temporal sliding-window weight gradient apply
    z = relu(y * scale + shift) to their LDS windows instead of reading a materialised z -- bit-identical outputs and
    statistics (same bf16 z values, same products, same order), the weight gradient equal up to its fp32 atomics."""
    Ci, Co, B, T, Hh, W = case
    k, s, p = (3, 1, 1), (1, 1, 1), (1, 0, 0)
    g = torch.Generator().manual_seed(5 * Ci + T + W)
    geom = L.ConvGeom(B, (T, Hh, W), Ci, Co, k, s, p)
    try:
        H.set_option("sw_wgrad", 1)         # (small test shapes: take the temporal kernel wherever it is built)
        assert L.can_fuse_bn_apply(geom), "this shape should be one the fused path takes"
        y = torch.randn(geom.Min, geom.in_cstride, generator=g).to(torch.bfloat16).to(DEV)
        dy = torch.randn(geom.M, geom.out_cstride, generator=g).to(torch.bfloat16).to(DEV)
        y[:, Ci:] = 0
        scale = (0.5 + torch.rand(geom.in_cstride, generator=g)).to(DEV)
        shift = (0.3 * torch.randn(geom.in_cstride, generator=g)).to(DEV)
        scale[Ci:] = 0      # (as pp_bn_finalize leaves the pad channels)
        shift[Ci:] = 0
        wf, _ = L.prep_conv_weights((torch.randn(Co, Ci, *k, generator=g) / math.sqrt(3 * Ci)).to(DEV), geom)
        for relu in (True, False):
            z = torch.empty_like(y)
            H.bn_apply(y, scale, shift, None, relu, z, geom.Min, geom.in_cstride)
            o_ref, st_ref = L.conv_fwd(z, geom, wf, stats=True)
            o, st = L.conv_fwd(y, geom, wf, stats=True, x_bn=(scale, shift, relu))
            gw_ref = L.conv_wgrad_raw(z, dy, geom)
            gw = L.conv_wgrad_raw(y, dy, geom, x_bn=(scale, shift, relu))
            torch.cuda.synchronize()
            assert torch.equal(o, o_ref) and torch.equal(st, st_ref), relu
            sc = gw_ref.abs().max().item()
            assert (gw - gw_ref).abs().max().item() <= 2e-5 * sc + 1e-6, relu
    finally:
        H.set_option("sw_wgrad", H.SW_WGRAD_DEFAULT)
    # a shape neither kernel takes says so, and asking anyway fails loudly instead of ignoring the parameters
    other = L.ConvGeom(2, (4, 6, 6), 32, 48, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    assert not L.can_fuse_bn_apply(other)
    wo, _ = L.prep_conv_weights(torch.randn(48, 32, 1, 3, 3, device=DEV), other)
    xo = torch.randn(other.Min, other.in_cstride, device=DEV).to(torch.bfloat16)
    with pytest.raises(Exception):
        L.conv_fwd(xo, other, wo, x_bn=(torch.ones(other.in_cstride, device=DEV), torch.zeros(other.in_cstride, device=DEV), True))


@pytest.mark.parametrize("case", [
    # Co, k, s, p, B, T, H, W   -- first convolutions over three input channels with stride 2 along W
    (45, (1, 7, 7), (1, 2, 2), (0, 3, 3), 2, 3, 20, 24),      # r2plus1d_18's stem
    (64, (3, 7, 7), (1, 2, 2), (1, 3, 3), 1, 4, 12, 16),      # r3d_18 / mc3_18
    (64, (1, 7, 7), (1, 2, 2), (0, 3, 3), 3, 1, 16, 18),      # resnet18 (frames as a batch of images)
    (16, (1, 3, 3), (1, 2, 2), (0, 1, 1), 1, 2, 9, 10),       # even padding: two pair taps
])
def test_paired_pixel_stem_matches_conv3d(case):
    """The first convolution over an input stored with four channels per pixel, run as a stride-1 convolution over pixel
    pairs (ConvGeom.paired_stem, pp_video_normalize_ndhwc4, pp_prep_conv_weight_pairs / pp_unprep_conv_grad_pairs): forward
    and weight gradient against torch's conv3d on the same bf16-rounded operands, and bit-identical to the 8-channel route."""
    Co, k, s, p, B, T, Hh, W = case
    g = torch.Generator().manual_seed(Co + W)
    x = torch.rand(B, 3, T, Hh, W, generator=g)
    w = (torch.randn(Co, 3, *k, generator=g) / math.sqrt(3 * k[0] * k[1] * k[2]))
    mean, std = (0.3, 0.4, 0.5), (0.2, 0.25, 0.3)
    geom = L.ConvGeom.paired_stem(B, (T, Hh, W), 3, Co, k, s, p)
    plain = L.ConvGeom(B, (T, Hh, W), 3, Co, k, s, p, in_cstride=8, cg_in=8)
    assert geom is not None and geom.out_thw == plain.out_thw and geom.Kf < plain.Kf
    x4 = torch.empty(B * T * Hh * W, 4, dtype=torch.bfloat16, device=DEV)
    x8 = torch.empty(B * T * Hh * W, 8, dtype=torch.bfloat16, device=DEV)
    H.video_normalize_ndhwc(x.to(DEV), x4, mean, std)
    H.video_normalize_ndhwc(x.to(DEV), x8, mean, std)
    assert torch.equal(x4[:, :3], x8[:, :3]) and (x4[:, 3] == 0).all()
    wf, _ = L.prep_conv_weights(w.to(DEV), geom, need_dgrad=False)
    wf8, _ = L.prep_conv_weights(w.to(DEV), plain, need_dgrad=False)
    y, st = L.conv_fwd(x4, geom, wf, stats=True)
    y8, st8 = L.conv_fwd(x8, plain, wf8, stats=True)
    dyf = rb(torch.randn(B, Co, *geom.out_thw, generator=g))
    dy = to_cl(dyf, geom.out_cstride)
    dw = L.conv_wgrad(x4, dy, geom, tuple(w.shape))
    dw8 = L.conv_wgrad(x8, dy, plain, tuple(w.shape))
    torch.cuda.synchronize()
    # same products in a different order of summation (K is walked pair by pair): bf16 outputs one rounding step apart at most
    scale = y8.float().abs().max().item()
    assert (y.float() - y8.float()).abs().max().item() <= 2.0 ** -7 * scale
    xn = x8[:, :3].float().cpu().view(B, T, Hh, W, 3).permute(0, 4, 1, 2, 3)
    wr = rb(w).requires_grad_(True)
    ref = F.conv3d(xn, wr, stride=s, padding=p)
    close(from_cl(y, B, geom.out_thw, Co), ref, name="paired stem forward vs conv3d")
    ref.backward(dyf)
    close(dw, wr.grad, name="paired stem weight gradient vs conv3d")
    sc = dw8.abs().max().item()
    assert (dw - dw8).abs().max().item() <= 2e-5 * sc + 1e-6


def test_one_launch_weight_preparation_matches_the_per_convolution_launches():
    """layers.PrepPlan (pp_prep_conv_weight_multi): a tower's convolution operands built in one launch on the second and
    later passes are bit-identical to pp_prep_conv_weight's, also after the weights have moved, for both layouts, channel
    counts that are no multiple of 8 / 64 (45, 230, 460) and convolutions without a data gradient."""
    torch.manual_seed(3)
    shapes = [(45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 8, 8), True), (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 8, 8), True),
              (64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 8, 8), True), (230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (4, 4, 4), True),
              (256, 460, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 4, 4), True), (512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (33, 1, 1), False)]
    ws = [torch.randn(Co, Ci, *k, device="cuda") * 0.1 for (Ci, Co, k, _, _, _, _) in shapes]
    geoms = [L.ConvGeom(2, thw, Ci, Co, k, st, pd) for (Ci, Co, k, st, pd, thw, _) in shapes]

    def one_pass():
        with L.PrepPlan(("test-prep-plan",)) as plan:
            got = [L.prep_conv_weights(w, g, need_dgrad=sh[6]) for w, g, sh in zip(ws, geoms, shapes)]
            return got, len(plan.ready)

    L.PrepPlan._plans.pop(("test-prep-plan",), None)
    first, left = one_pass()                       # records; every operand built on its own
    assert left == 0
    for it in range(2):
        for w in ws:
            w.mul_(1.5).add_(0.01)                 # the optimizer moves the masters
        got, left = one_pass()                     # built on entry, in one launch
        assert left == 0, "every prepared operand was asked for"
        prev, L.PREP_PLAN = L.PREP_PLAN, False
        try:
            ref, _ = one_pass()
        finally:
            L.PREP_PLAN = prev
        for (wf, wd), (rf, rd), sh in zip(got, ref, shapes):
            assert torch.equal(wf, rf)
            assert (wd is None and rd is None) if not sh[6] else torch.equal(wd, rd)
    L.PrepPlan._plans.pop(("test-prep-plan",), None)


@pytest.mark.parametrize("case", [
    ("dense", 1000, 768, 768), ("dense", 7296, 2304, 768), ("dense", 4101, 520, 3072), ("dense", 333, 200, 264),
    ("conv", 2, 512, 512, (3, 1, 1), (2, 1, 1), (0, 0, 0), (1001, 1, 1)),          # wav2vec2 feature extractor, k = 3 stride 2
    ("conv", 3, 512, 512, (2, 1, 1), (2, 1, 1), (0, 0, 0), (459, 1, 1)),
    ("conv", 2, 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 28, 28)),             # layer 2.0 strided spatial (padding, ragged Ni)
    ("conv", 2, 256, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 14, 14)),
    ("conv", 2, 576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 7, 9)),
])
def test_wide_tile_weight_gradient_matches_the_128_tile_kernel_and_torch(case):
    """pp_wgrad's 256 x 256 tiles (wgrad_big_kernel, round 4: half the L2 traffic of the 128 x 128 tiles) against the
    128 x 128 kernel on the same operands (fp32 summation order only: the M splits differ) and, for the dense cases, against
    torch in fp32; the grouped launch with the fused bias gradient is bitwise the 128-tile one (no M split in either)."""
    torch.manual_seed(5)
    H.set_option("sw_wgrad", 0)            # (the sliding-window kernels would take the stride-1 convolutions)
    try:
        if case[0] == "dense":
            _, M, N, K = case
            x = torch.randn(M, K, device="cuda").bfloat16()
            dy = torch.randn(M, N, device="cuda").bfloat16()
            H.set_option("wgrad_big", 0)
            ref, rb = [t.clone() for t in L.linear_wgrad(x, dy, M, N, K)]
            H.set_option("wgrad_big", 1)
            got, gb = [t.clone() for t in L.linear_wgrad(x, dy, M, N, K)]
            want = dy.float().t() @ x.float()
            scale = want.abs().max().item()
            assert (got - ref).abs().max().item() <= 2e-5 * scale
            assert (got - want).abs().max().item() <= 2e-5 * scale
            assert (gb - rb).abs().max().item() <= 2e-5 * rb.abs().max().item()
            assert (gb - dy.float().sum(0)).abs().max().item() <= 2e-5 * rb.abs().max().item() + 1e-3
            if M >= 4096:           # grouped launches: no M split, bitwise equal including the bias gradient
                items = [(torch.randn(M, K, device="cuda").bfloat16(), torch.randn(M, N, device="cuda").bfloat16()) for _ in range(3)]
                big = [(a.clone(), b.clone()) for a, b in L.linear_wgrad_group(items, M, N, K)]
                H.set_option("wgrad_big", 0)
                small = [(a.clone(), b.clone()) for a, b in L.linear_wgrad_group(items, M, N, K)]
                for (a, b), (c, d) in zip(big, small):
                    assert torch.equal(a, c) and torch.equal(b, d)
        else:
            _, B, Ci, Co, k, st, pd, thw = case
            geom = L.ConvGeom(B, thw, Ci, Co, k, st, pd)
            x = torch.randn(geom.Min, geom.in_cstride, device="cuda").bfloat16()
            dy = torch.randn(geom.M, geom.out_cstride, device="cuda").bfloat16()
            H.set_option("wgrad_big", 0)
            ref = L.conv_wgrad_raw(x, dy, geom).clone()
            H.set_option("wgrad_big", 1)
            got = L.conv_wgrad_raw(x, dy, geom).clone()
            assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    finally:
        H.set_option("wgrad_big", 32768)
        H.set_option("sw_wgrad", 4096)


def test_wide_tile_forward_and_data_gradient_are_bit_identical_to_the_default_kernels():
    """pp_igemm's 256 x 256 tiles (igemm_big_kernel, pp_set_option igemm_big; off by default -- DESIGN.md section 8): the same
    K order and the same epilogue roundings as the ring / gather kernels, so the results are the same bits: strided
    convolution forward with GELU + the saved pre-activation, the parity classes of a strided data gradient (output row map),
    a padded 3-D strided convolution, and a dense GEMM with bias / GELU / dropout / residual, ragged M, N and K."""
    torch.manual_seed(11)

    def both(fn):
        H.set_option("igemm_big", 0)
        ref = fn()
        H.set_option("igemm_big", -1)          # (negative: fused epilogues too; |value| = minimum M)
        try:
            got = fn()
        finally:
            H.set_option("igemm_big", 0)
        return ref, got

    for B, k, T in ((5, 3, 9999), (7, 2, 7001)):
        geom = L.ConvGeom(B, (T, 1, 1), 512, 512, (k, 1, 1), (2, 1, 1), (0, 0, 0))
        x = torch.randn(geom.Min, 512, device="cuda").bfloat16()
        dy = torch.randn(geom.M, 512, device="cuda").bfloat16()
        w = torch.randn(512, 512, k, 1, 1, device="cuda") * 0.03
        wf, wd = L.prep_conv_weights(w, geom)

        def fwd():
            pre = L.empty((geom.M, 512), torch.bfloat16, x)
            y, _ = L.conv_fwd(x, geom, wf, act=H.ACT_GELU, pre=pre)
            return y.clone(), pre.clone()
        (y0, p0), (y1, p1) = both(fwd)
        assert torch.equal(y0, y1) and torch.equal(p0, p1)
        assert y0.float().abs().max().item() > 0.1
        d0, d1 = both(lambda: L.conv_dgrad(dy, geom, wd).clone())
        assert torch.equal(d0, d1)
    geom = L.ConvGeom(8, (8, 56, 56), 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x = torch.randn(geom.Min, geom.in_cstride, device="cuda").bfloat16()
    wf, _ = L.prep_conv_weights(torch.randn(230, 64, 1, 3, 3, device="cuda") * 0.05, geom)
    y0, y1 = both(lambda: L.conv_fwd(x, geom, wf)[0].clone())
    assert torch.equal(y0, y1)
    M, N, K = 50011, 776, 520
    wl = torch.randn(N, K, device="cuda") * 0.05
    wfl, _ = L.prep_linear(wl)
    xp = torch.zeros(M, wfl.shape[1], device="cuda", dtype=torch.bfloat16)
    xp[:, :K] = torch.randn(M, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    Np = L.cpad(N)
    res = torch.randn(M, Np, device="cuda").bfloat16()
    for kw in (dict(), dict(bias=bias, act=H.ACT_GELU), dict(bias=bias, residual=res, dropout=(0.1, 1234))):
        def lin():
            pre = L.empty((M, Np), torch.bfloat16, xp) if "act" in kw else None
            y = L.linear_fwd(xp, M, wfl, N, pre=pre, **kw)
            return y.clone(), (pre.clone() if pre is not None else None)
        (a0, q0), (a1, q1) = both(lin)
        Ns = (N + 7) // 8 * 8          # (the kernels store whole 8-column chunks; the pad columns beyond are never written)
        assert torch.equal(a0[:, :Ns], a1[:, :Ns]) and (q0 is None or torch.equal(q0[:, :Ns], q1[:, :Ns])), sorted(kw)


@pytest.mark.parametrize("B,T,Hh,W,Co", [(2, 3, 112, 112, 45), (1, 2, 64, 64, 45), (3, 1, 30, 50, 45), (2, 2, 112, 112, 48), (1, 1, 14, 128, 33)])
def test_stem_window_kernel_matches_the_gather_kernel_and_conv3d(B, T, Hh, W, Co):
    """pp_stem_pairs_fwd (round 4: the paired-pixel stem as a window kernel, the input read once): the same bits in y as
    pp_igemm's gather kernel on the same operands (same K order), column statistics equal to fp32 rounding (other partial
    rows), and y against torch's Conv3d in fp32; odd sizes (Ho not a multiple of 4, Wp not a multiple of 16, the widest
    frame) and 48 / 33 output channels."""
    torch.manual_seed(2)
    geom = L.ConvGeom.paired_stem(B, (T, Hh, W), 3, Co, (1, 7, 7), (1, 2, 2), (0, 3, 3))
    assert geom is not None and L._stem_window_ok(geom)
    xv = torch.rand(B, 3, T, Hh, W, device="cuda")
    x4 = torch.zeros(B, T, Hh, W, 4, device="cuda")
    x4[..., :3] = xv.permute(0, 2, 3, 4, 1)
    x = x4.bfloat16().view(-1, 8)
    w = torch.randn(Co, 3, 1, 7, 7, device="cuda") * 0.05
    wf, _ = L.prep_conv_weights(w, geom, need_dgrad=False)
    prev = L.STEM_WINDOW
    try:
        L.STEM_WINDOW = False
        y0, p0 = L.conv_fwd(x, geom, wf, stats=True)
        L.STEM_WINDOW = True
        y1, p1 = L.conv_fwd(x, geom, wf, stats=True)
        y2, none = L.conv_fwd(x, geom, wf)
    finally:
        L.STEM_WINDOW = prev
    nc = (Co + 7) // 8 * 8
    assert torch.equal(y0[:, :nc], y1[:, :nc]) and torch.equal(y1[:, :nc], y2[:, :nc]) and none is None
    s0, s1 = p0.double().sum(0)[:, :Co], p1.double().sum(0)[:, :Co]
    assert (s0 - s1).abs().max().item() <= 1e-6 * s0.abs().max().item()
    ref = F.conv3d(xv.bfloat16().float(), w.bfloat16().float(), stride=(1, 2, 2), padding=(0, 3, 3))      # (B, Co, T, Ho, Wo)
    got = y1[:, :Co].float().view(B, T, geom.Ho, geom.Wo, Co).permute(0, 4, 1, 2, 3)
    assert (got - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-3
    # the weight gradient on the same window (pp_stem_pairs_wgrad): against the gather kernel (fp32 summation order) and torch
    dy = torch.zeros(geom.M, geom.out_cstride, device="cuda", dtype=torch.bfloat16)
    dy[:, :Co] = torch.randn(geom.M, Co, device="cuda").bfloat16()
    try:
        L.STEM_WINDOW = False
        g0 = L.conv_wgrad(x, dy, geom, w.shape).clone()
        L.STEM_WINDOW = True
        g1 = L.conv_wgrad(x, dy, geom, w.shape).clone()
    finally:
        L.STEM_WINDOW = prev
    assert (g1 - g0).abs().max().item() <= 2e-5 * g0.abs().max().item()
    xr = xv.bfloat16().float().requires_grad_(False)
    wr = w.clone().requires_grad_(True)
    out = F.conv3d(xr, wr, stride=(1, 2, 2), padding=(0, 3, 3))
    out.backward(dy[:, :Co].float().view(B, T, geom.Ho, geom.Wo, Co).permute(0, 4, 1, 2, 3).contiguous())
    assert (g1 - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item()
