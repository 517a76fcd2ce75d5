"""wav2vec2-base on the HIP path.

Module tree and parameter names equal torchaudio 0.9.1 `models.wav2vec2` (what
`pig/models.py:70-74` builds via `A.wav2vec2_base(num_out=28)`): `feature_extractor.conv_layers.N.
{conv,layer_norm}`, `encoder.feature_projection.{layer_norm,projection}`, `encoder.transformer.
{pos_conv_embed.conv.{weight_g,weight_v,bias}, layer_norm, layers.N.{attention.{k,v,q,out}_proj,
layer_norm, feed_forward.{intermediate,output}_dense, final_layer_norm}}`, `encoder.readout`.
The torch.nn modules are parameter containers; forward/backward are hand-scheduled chains of HIP
kernels (strided-GEMM conv stack, grouped positional conv, MFMA GEMMs with bias/GELU/residual
epilogues, LayerNorm, batched attention) driven by `Wav2Vec2Fn`.

Dropout / LayerDrop follow the modules' `p` / `layer_drop` in train mode (torchaudio defaults 0.1): masks
come from a counter-based hash (seed, element index) and are regenerated in the backward pass; LayerDrop
uses a host RNG like torchaudio (`torch.rand(1).item() <= layer_drop`), seeded identically on every rank.  Parity tests set p = 0.
"""
import torch
from torch import nn

from . import hip as H
from . import layers as L
from .hip import act16, f32
from .dist import grad_dict

CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
NUM_HEADS = 12


def n_frames(length):
    for _, k, s in CONV_SPEC:
        length = (length - k) // s + 1
    return length


class ConvLayerBlock(nn.Module):
    def __init__(self, ci, co, k, s, norm):
        super().__init__()
        self.conv = nn.Conv1d(ci, co, k, s, bias=False)
        self.layer_norm = nn.GroupNorm(co, co, affine=True) if norm else None


class FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        ci, blocks = 1, []
        for i, (co, k, s) in enumerate(CONV_SPEC):
            blocks.append(ConvLayerBlock(ci, co, k, s, i == 0))
            ci = co
        self.conv_layers = nn.ModuleList(blocks)

    def forward(self, x, length=None):
        """(B, L) -> ((B, T, 512), None); the `feature_extractor(x, None)` call of pig/grsa.py:448-452."""
        return _features_only(self, x), length


class FeatureProjection(nn.Module):
    def __init__(self, ci, co, p):
        super().__init__()
        self.layer_norm = nn.LayerNorm(ci)
        self.projection = nn.Linear(ci, co)
        self.dropout = nn.Dropout(p)


class ConvolutionalPositionalEmbedding(nn.Module):
    def __init__(self, dim=768, kernel=128, groups=16):
        super().__init__()
        self.embed_dim, self.kernel, self.groups = dim, kernel, groups
        self.conv = nn.Conv1d(dim, dim, kernel, padding=kernel // 2, groups=groups)
        self.conv = torch.nn.utils.weight_norm(self.conv, name="weight", dim=2)
        self.num_remove = 1 if kernel % 2 == 0 else 0


class SelfAttention(nn.Module):
    def __init__(self, dim, heads, p):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = dim, heads, dim // heads
        self.scaling = self.head_dim ** -0.5
        self.dropout = nn.Dropout(p)
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)


class FeedForward(nn.Module):
    def __init__(self, dim, inner, p1, p2):
        super().__init__()
        self.intermediate_dense = nn.Linear(dim, inner)
        self.intermediate_dropout = nn.Dropout(p1)
        self.output_dense = nn.Linear(inner, dim)
        self.output_dropout = nn.Dropout(p2)


class EncoderLayer(nn.Module):
    def __init__(self, dim, heads, inner, p):
        super().__init__()
        self.attention = SelfAttention(dim, heads, p)
        self.dropout = nn.Dropout(p)
        self.layer_norm = nn.LayerNorm(dim)
        self.layer_norm_first = False
        self.feed_forward = FeedForward(dim, inner, p, p)
        self.final_layer_norm = nn.LayerNorm(dim)


class Transformer(nn.Module):
    def __init__(self, dim, heads, inner, n_layers, p, layer_drop):
        super().__init__()
        self.pos_conv_embed = ConvolutionalPositionalEmbedding(dim)
        self.layer_norm = nn.LayerNorm(dim)
        self.layer_norm_first = True  # torchaudio's flag is the negation of the fairseq config
        self.layer_drop = layer_drop
        self.dropout = nn.Dropout(p)
        self.layers = nn.ModuleList([EncoderLayer(dim, heads, inner, p) for _ in range(n_layers)])
        # LayerDrop decisions come from a dedicated host generator with a fixed seed, so that all data-parallel
        # ranks drop the same layers (a parameter then has a gradient on every rank or on none)
        self._ld_gen = torch.Generator().manual_seed(0x5EED)


class Encoder(nn.Module):
    def __init__(self, num_out, p, layer_drop):
        super().__init__()
        self.feature_projection = FeatureProjection(512, 768, p)
        self.transformer = Transformer(768, NUM_HEADS, 3072, 12, p, layer_drop)
        self.readout = nn.Linear(768, num_out)


class Wav2Vec2Model(nn.Module):
    """`forward(wave (B,L)) -> ((B,T,num_out) fp32, None)`, `extract_features -> ((B,T,512), None)`."""

    def __init__(self, num_out=28, dropout=0.1, layer_drop=0.1):
        super().__init__()
        self.num_out = num_out
        self.feature_extractor = FeatureExtractor()
        self.encoder = Encoder(num_out, dropout, layer_drop)

    def extract_features(self, waveforms, lengths=None):
        return _features_only(self.feature_extractor, waveforms), lengths

    def forward(self, waveforms, lengths=None):
        if lengths is not None:
            raise NotImplementedError("lengths/masking is not used by the reference (pig/models.py:103)")
        ps = [p for p in self.parameters()]
        return Wav2Vec2Fn.apply(waveforms, self, True, torch.is_grad_enabled(), *ps), None


def wav2vec2_base(num_out, dropout=0.1, layer_drop=0.1):
    return Wav2Vec2Model(num_out, dropout, layer_drop)


def _features_only(fe, wave):
    model = _Holder(fe)
    ps = [p for p in fe.parameters()]
    return Wav2Vec2Fn.apply(wave, model, False, torch.is_grad_enabled(), *ps)


class _Holder:
    """Lets Wav2Vec2Fn run the feature extractor alone (extract_features)."""

    def __init__(self, fe):
        self.feature_extractor = fe
        self.encoder = None

    def parameters(self):
        return self.feature_extractor.parameters()


class _Rec:
    pass


class _Drop:
    """Per-forward dropout bookkeeping: one fresh seed per dropout site."""

    def __init__(self, training):
        self.on = training
        self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if training else 0
        self.n = 0

    def site(self, module):
        """(p, seed) for one dropout site; p = 0 when not training."""
        p = float(module.p) if (self.on and module is not None) else 0.0
        self.n += 1
        return p, (self.base + self.n * 0x9E3779B1) & 0xffffffff


# ---------------------------------------------------------------------------------------------------
def _fe_forward(fe, wave, save):
    """wave fp32 [B][L] -> features 16-bit [B*T][512]; tape for backward."""
    B, Lw = wave.shape
    t = _Rec()
    t.B, t.L = B, Lw
    blk0 = fe.conv_layers[0]
    T0 = (Lw - 10) // 5 + 1
    t.T0 = T0
    t.w0 = blk0.conv.weight.reshape(512, 10)
    t.stats = L.zeros((B, 512, 2), f32, wave)
    H.conv0_stats(wave, B, Lw, T0, t.w0, t.stats)
    cur = L.empty((B * T0, 512), act16(), wave)
    H.conv0_apply(wave, B, Lw, T0, t.w0, t.stats, blk0.layer_norm.weight, blk0.layer_norm.bias, blk0.layer_norm.eps, cur)
    t.wave, t.layers = wave, []
    Tcur = T0
    with L.PrepPlan(("wav2vec2-fe", id(fe), (B, Lw), bool(save))):       # the six convolutions' operands in one launch
        for blk in fe.conv_layers[1:]:
            k, s = blk.conv.kernel_size[0], blk.conv.stride[0]
            geom = L.ConvGeom(B, (Tcur, 1, 1), 512, 512, (k, 1, 1), (s, 1, 1), (0, 0, 0))
            wf, wd = L.prep_conv_weights(blk.conv.weight, geom, need_dgrad=save)
            pre = L.empty((geom.M, 512), act16(), wave) if save else None
            y, _ = L.conv_fwd(cur, geom, wf, act=H.ACT_GELU, pre=pre)
            r = _Rec()
            r.blk, r.geom, r.wd, r.x, r.pre = blk, geom, wd, cur, pre
            t.layers.append(r)
            cur, Tcur = y, geom.To
    t.T = Tcur
    return cur, Tcur, t


def _fe_backward(fe, t, dfeat, grads):
    cur = dfeat
    for r in reversed(t.layers):
        du = L.empty(cur.shape, act16(), cur)
        H.gelu_bwd(cur, r.pre, du)
        if r.blk.conv.weight.requires_grad:
            grads[r.blk.conv.weight] = L.conv_wgrad(r.x, du, r.geom, r.blk.conv.weight.shape)
        cur = L.conv_dgrad(du, r.geom, r.wd)
    blk0 = fe.conv_layers[0]
    if blk0.conv.weight.requires_grad:
        gn = blk0.layer_norm
        red = L.zeros((t.B, 512, 2), f32, cur)
        dw, dg, db = L.zeros((512, 10), f32, cur), L.zeros((512,), f32, cur), L.zeros((512,), f32, cur)
        H.conv0_bwd_reduce(t.wave, t.B, t.L, t.T0, t.w0, t.stats, gn.weight, gn.bias, gn.eps, cur, red)
        H.conv0_bwd_apply(t.wave, t.B, t.L, t.T0, t.w0, t.stats, gn.weight, gn.bias, gn.eps, cur, red, dw, dg, db)
        grads[blk0.conv.weight] = dw.view(512, 1, 10)
        grads[gn.weight], grads[gn.bias] = dg, db


def _prep_qkv(att, need_dgrad, batch):
    """q/k/v projections fused into one [2304][768] operand (+ its transpose) and one bias vector (jobs of `batch`)."""
    lins = (att.q_proj, att.k_proj, att.v_proj)
    w0 = lins[0].weight
    wf = L.empty((2304, 768), act16(), w0)
    wt = L.empty((768, 2304), act16(), w0) if need_dgrad else None
    bias = L.empty((2304,), f32, w0)
    for i, lin in enumerate(lins):
        batch.cast(lin.weight, wf[i * 768:], 768, 768, 768, 768, 768, 768)
        if need_dgrad:
            batch.cast(lin.weight, wt[:, i * 768:], 768, 768, 768, 768, 768, 2304, transpose=True)
        batch.cast(lin.bias, bias[i * 768:], 1, 768, 768, 1, 768, 768, out_f32=True)
    return wf, wt, bias


def _prep_encoder_weights(enc, save):
    """Every Linear operand of the encoder (16-bit [N][K] and, for the backward pass, its transpose) in ONE launch;
    kept until a master changes (layers.cached_operands: gradient accumulation, validation, forward-only loops)."""
    fp, tr = enc.feature_projection, enc.transformer
    params = [fp.projection.weight, enc.readout.weight]
    for layer in tr.layers:
        att, ff = layer.attention, layer.feed_forward
        for lin in (att.q_proj, att.k_proj, att.v_proj):
            params += [lin.weight, lin.bias]
        params += [att.out_proj.weight, ff.intermediate_dense.weight, ff.output_dense.weight]
    return L.cached_operands(("wav2vec2-encoder", bool(save)), params, lambda: _build_encoder_weights(enc, save))


def _build_encoder_weights(enc, save):
    fp, tr = enc.feature_projection, enc.transformer
    batch = L.CastBatch()
    w = {"proj": batch.linear(fp.projection.weight, save), "readout": batch.linear(enc.readout.weight, save)}
    for layer in tr.layers:
        att, ff = layer.attention, layer.feed_forward
        w[layer] = (_prep_qkv(att, save, batch), batch.linear(att.out_proj.weight, save),
                    batch.linear(ff.intermediate_dense.weight, save), batch.linear(ff.output_dense.weight, save))
    batch.run(fp.projection.weight.device)
    return w, batch


FUSED_ATTENTION = True   # one launch per direction (pp_attention_*) when T <= 320; False: batched GEMMs + softmax + transposes
FUSED_ATTENTION_MAX_T = 320     # (316 frames = the reference's own 2.3-s clips at 44.1 kHz; longer clips take the unfused path, T <= 1024)


def _attention_fwd(qkv, B, T, Tp, scale, save, drop=(0.0, 0)):
    """qkv 16-bit [B*T][2304] -> ctx 16-bit [B*T][768]; returns what the backward pass needs besides qkv and ctx: the saved P
    (before attention dropout) on the unfused path, the rows' log-sum-exp (fp32 [B*12][T]) on the fused one, which
    recomputes the probabilities."""
    Hn, Dh, D3 = NUM_HEADS, 64, 2304
    if FUSED_ATTENTION and T <= FUSED_ATTENTION_MAX_T:
        ctx = L.empty((B * T, 768), act16(), qkv)
        lse = L.empty((B * Hn, T), f32, qkv)
        H.attention_fwd(qkv, B, T, Hn, scale, drop[0], drop[1], ctx, lse)
        return ctx, lse
    nb = B * Hn
    S = L.empty((nb, T, Tp), f32, qkv)
    q, k, v = qkv, qkv[:, 768:], qkv[:, 1536:]   # column views (pointer offsets only)
    H.igemm(q, k, S, T, T, Dh, H.gather_dense(D3), D3, Tp, nbatch=nb, inner=Hn, a_s=(T * D3, Dh), b_s=(T * D3, Dh),
            c_s=(Hn * T * Tp, T * Tp))
    P = L.empty((nb, T, Tp), act16(), qkv)
    H.softmax_fwd(S, Tp, P, Tp, nb, T, scale)
    Vt = L.empty((nb, Dh, Tp), act16(), qkv)
    H.transpose_bf16(v, T * D3, D3, Vt, Hn * Dh * Tp, Tp, nb, T, Dh, inner=Hn, in_s1=Dh, out_s1=Dh * Tp)
    Pd = P
    if drop[0] > 0:
        Pd = L.empty(P.shape, act16(), qkv)
        H.dropout_bf16(P, Pd, drop[0], drop[1])
    ctx = L.empty((B * T, 768), act16(), qkv)
    H.igemm(Pd, Vt, ctx, T, Dh, Tp, H.gather_dense(Tp), Tp, 768, nbatch=nb, inner=Hn, a_s=(Hn * T * Tp, T * Tp),
            b_s=(Hn * Dh * Tp, Dh * Tp), c_s=(T * 768, Dh))
    return ctx, (P if save else None)


def _attention_bwd(dctx, qkv, ctx, P, B, T, Tp, scale, drop=(0.0, 0)):
    """-> dqkv 16-bit [B*T][2304].  P: what _attention_fwd returned (fp32 log-sum-exp rows = the fused path)."""
    Hn, Dh, D3 = NUM_HEADS, 64, 2304
    if P is not None and P.dtype == f32:   # forward ran fused
        dqkv = L.empty((B * T, D3), act16(), qkv)
        H.attention_bwd(qkv, ctx, P, dctx, B, T, Hn, scale, drop[0], drop[1], dqkv)
        return dqkv
    nb = B * Hn
    q, k, v = qkv, qkv[:, 768:], qkv[:, 1536:]
    dqkv = L.empty((B * T, D3), act16(), qkv)
    dq, dk, dv = dqkv, dqkv[:, 768:], dqkv[:, 1536:]
    # dP = dctx V^T
    dP = L.empty((nb, T, Tp), f32, qkv)
    H.igemm(dctx, v, dP, T, T, Dh, H.gather_dense(768), D3, Tp, nbatch=nb, inner=Hn, a_s=(T * 768, Dh),
            b_s=(T * D3, Dh), c_s=(Hn * T * Tp, T * Tp))
    Pd = P
    if drop[0] > 0:   # same mask as the forward pass: dP through the dropout, dV from the dropped probabilities
        H.dropout_f32(dP, dP, drop[0], drop[1])
        Pd = L.empty(P.shape, act16(), qkv)
        H.dropout_bf16(P, Pd, drop[0], drop[1])
    # dV = Pd^T dctx
    Pt = L.empty((nb, T, Tp), act16(), qkv)
    H.transpose_bf16(Pd, T * Tp, Tp, Pt, T * Tp, Tp, nb, T, T)
    dOt = L.empty((nb, Dh, Tp), act16(), qkv)
    H.transpose_bf16(dctx, T * 768, 768, dOt, Hn * Dh * Tp, Tp, nb, T, Dh, inner=Hn, in_s1=Dh, out_s1=Dh * Tp)
    H.igemm(Pt, dOt, dv, T, Dh, Tp, H.gather_dense(Tp), Tp, D3, nbatch=nb, inner=Hn, a_s=(Hn * T * Tp, T * Tp),
            b_s=(Hn * Dh * Tp, Dh * Tp), c_s=(T * D3, Dh))
    # dS = softmax'(P, dP)
    dS = L.empty((nb, T, Tp), act16(), qkv)
    H.softmax_bwd(dP, Tp, P, Tp, dS, nb, T, scale)
    # dQ = dS K
    Kt = L.empty((nb, Dh, Tp), act16(), qkv)
    H.transpose_bf16(k, T * D3, D3, Kt, Hn * Dh * Tp, Tp, nb, T, Dh, inner=Hn, in_s1=Dh, out_s1=Dh * Tp)
    H.igemm(dS, Kt, dq, T, Dh, Tp, H.gather_dense(Tp), Tp, D3, nbatch=nb, inner=Hn, a_s=(Hn * T * Tp, T * Tp),
            b_s=(Hn * Dh * Tp, Dh * Tp), c_s=(T * D3, Dh))
    # dK = dS^T Q
    dSt = L.empty((nb, T, Tp), act16(), qkv)
    H.transpose_bf16(dS, T * Tp, Tp, dSt, T * Tp, Tp, nb, T, T)
    Qt = L.empty((nb, Dh, Tp), act16(), qkv)
    H.transpose_bf16(q, T * D3, D3, Qt, Hn * Dh * Tp, Tp, nb, T, Dh, inner=Hn, in_s1=Dh, out_s1=Dh * Tp)
    H.igemm(dSt, Qt, dk, T, Dh, Tp, H.gather_dense(Tp), Tp, D3, nbatch=nb, inner=Hn, a_s=(Hn * T * Tp, T * Tp),
            b_s=(Hn * Dh * Tp, Dh * Tp), c_s=(T * D3, Dh))
    return dqkv


def _enc_forward(enc, feat, B, T, save, training=False):
    """feat 16-bit [B*T][512] -> out fp32 [B*T][num_out]; tape."""
    M = B * T
    Tp = L.cpad(T)
    t = _Rec()
    t.B, t.T, t.Tp, t.M = B, T, Tp, M
    fp, tr = enc.feature_projection, enc.transformer
    t.feat = feat
    xln, t.ln0 = L.layernorm_fwd(feat, fp.layer_norm, fp.layer_norm.eps)
    t.xln = xln
    W, t.prep_batch = _prep_encoder_weights(enc, save)
    wf, t.proj_wt = W["proj"]
    x0 = L.linear_fwd(xln, M, wf, 768, bias=fp.projection.bias)
    drop = _Drop(training)
    t.d_fp = drop.site(fp.dropout)
    if t.d_fp[0] > 0:
        H.dropout_bf16(x0, x0, *t.d_fp)
    # positional conv (weight-normalised, grouped), GELU, + residual
    pc = tr.pos_conv_embed
    conv = pc.conv
    geom = L.ConvGeom(B, (T, 1, 1), 768, 768, (pc.kernel, 1, 1), (1, 1, 1), (pc.kernel // 2, 0, 0), groups=pc.groups, To=T)
    Cig = 768 // pc.groups
    t.wn_norm = L.empty((pc.kernel,), f32, feat)
    wfp = L.empty((768, pc.kernel, Cig), act16(), feat)
    H.weightnorm_fwd(conv.weight_v, conv.weight_g.reshape(-1), 768, Cig, pc.kernel, t.wn_norm, wfp)
    t.pc_geom = geom
    t.pc_pre = L.empty((M, 768), act16(), feat) if save else None
    x1 = L.empty((M, 768), act16(), feat)
    H.igemm(x0, wfp, x1, M, geom.Cog, geom.Kf, geom.g_fwd(), geom.Kf, 768, b_rows=geom.Cog, bias=conv.bias,
            act=H.ACT_GELU, Cpre=t.pc_pre, residual=x0, ldr=768, nbatch=pc.groups, inner=1, a_s=(Cig, 0),
            b_s=(geom.Cog * geom.Kf, 0), c_s=(geom.Cog, 0), bias_s=(geom.Cog, 0))
    t.x0, t.x1, t.wfp = x0, x1, wfp
    x, t.ln1 = L.layernorm_fwd(x1, tr.layer_norm, tr.layer_norm.eps)
    t.d_tr = drop.site(tr.dropout)
    if t.d_tr[0] > 0:
        H.dropout_bf16(x, x, *t.d_tr)
    t.layers = []
    for layer in tr.layers:
        if training and tr.layer_drop > 0 and torch.rand(1, generator=tr._ld_gen).item() <= tr.layer_drop:
            continue   # LayerDrop: the layer is skipped for this step (host RNG, like torchaudio)
        r = _Rec()
        att, ff = layer.attention, layer.feed_forward
        r.d_att, r.d_out = drop.site(att.dropout), drop.site(layer.dropout)
        r.d_int, r.d_ffo = drop.site(ff.intermediate_dropout), drop.site(ff.output_dropout)
        (wf, r.qkv_wt, bqkv), w_out, w_ff1, w_ff2 = W[layer]
        r.x_in = x
        qkv = L.linear_fwd(x, M, wf, 2304, bias=bqkv)
        ctx, r.P = _attention_fwd(qkv, B, T, Tp, att.scaling, save, r.d_att)
        r.qkv, r.ctx = qkv, ctx
        wf, r.out_wt = w_out
        # dropout sits in the GEMM epilogues: s = dropout(x W^T + b) + residual in one kernel
        s1 = L.linear_fwd(ctx, M, wf, 768, bias=att.out_proj.bias, residual=x, dropout=r.d_out)
        xa, r.lnA = L.layernorm_fwd(s1, layer.layer_norm, layer.layer_norm.eps)
        r.s1, r.xa = s1, xa
        wf, r.ff1_wt = w_ff1
        r.u = L.empty((M, 3072), act16(), feat) if save else None
        h = L.linear_fwd(xa, M, wf, 3072, bias=ff.intermediate_dense.bias, act=H.ACT_GELU, pre=r.u, dropout=r.d_int)
        r.h = h
        wf, r.ff2_wt = w_ff2
        s2 = L.linear_fwd(h, M, wf, 768, bias=ff.output_dense.bias, residual=xa, dropout=r.d_ffo)
        x, r.lnB = L.layernorm_fwd(s2, layer.final_layer_norm, layer.final_layer_norm.eps)
        r.s2, r.layer = s2, layer
        t.layers.append(r)
    t.x_final = x
    n_out = enc.readout.out_features
    wf, t.ro_wt = W["readout"]
    out = L.empty((M, n_out), f32, feat)
    H.igemm(x, wf, out, M, n_out, 768, H.gather_dense(768), 768, n_out, b_rows=n_out, bias=enc.readout.bias)
    return out, t


GROUP_WGRAD = True    # the transformer layers' weight gradients as ONE launch per Linear shape (layers.linear_wgrad_group)


class _Deferred:
    """Weight gradients of the transformer layers, collected during the backward walk and launched as one grouped
    problem per Linear shape once the walk has left the layers.  They are off the critical path (nothing downstream reads
    a dW), so deferring costs nothing, and a grouped launch has 12 x the tiles of a per-layer one: every tile reduces
    its whole M (no split-M atomics; bitwise reproducible).  Each layer's operands stay alive until then (~100 MB per
    layer at batch 64: irrelevant against 288 GB)."""

    def __init__(self, M):
        self.M, self.kinds = M, {}

    def add(self, kind, lins, x, dy, N, K):
        self.kinds.setdefault((kind, N, K), []).append((lins, x, dy))

    def flush(self, grads):
        for (kind, N, K), entries in self.kinds.items():
            outs = L.linear_wgrad_group([(x, dy) for _, x, dy in entries], self.M, N, K)
            for (lins, _, _), (dw, db) in zip(entries, outs):
                n_each = N // len(lins)
                for i, lin in enumerate(lins):
                    if lin.weight.requires_grad:
                        grads[lin.weight] = dw[i * n_each:(i + 1) * n_each] if len(lins) > 1 else dw
                    if lin.bias is not None and lin.bias.requires_grad:
                        grads[lin.bias] = db[i * n_each:(i + 1) * n_each] if len(lins) > 1 else db
        self.kinds = {}


def _lin_grads(grads, lin, x, dy, M, N, K, deferred=None, kind=None):
    if deferred is not None and lin.weight.requires_grad and lin.bias is not None and lin.bias.requires_grad:
        deferred.add(kind, (lin,), x, dy, N, K)
        return
    need_w, need_b = lin.weight.requires_grad, lin.bias is not None and lin.bias.requires_grad
    if need_w or need_b:
        dw, db = L.linear_wgrad(x, dy, M, N, K, want_bias=need_b)
        if need_w:
            grads[lin.weight] = dw
        if need_b:
            grads[lin.bias] = db


def _ln_bwd(grads, ln, dy, x, saved):
    dx, dg, db = L.layernorm_bwd(dy, x, ln, saved)
    if ln.weight.requires_grad:
        grads[ln.weight], grads[ln.bias] = dg, db
    return dx


def _enc_backward(enc, t, dout, grads):
    """dout fp32 [M][num_out] -> dfeat 16-bit [M][512]."""
    M, B, T, Tp = t.M, t.B, t.T, t.Tp
    fp, tr = enc.feature_projection, enc.transformer
    n_out = enc.readout.out_features
    Np = L.cpad(n_out)
    dy = L.empty((M, Np), act16(), dout)
    H.cast_pad_2d(dout, dy, M, n_out, n_out, M, Np)
    _lin_grads(grads, enc.readout, t.x_final, dy, M, n_out, 768)
    dx = L.linear_dgrad(dy, M, t.ro_wt, 768)
    deferred = _Deferred(M) if GROUP_WGRAD else None
    for r in reversed(t.layers):
        layer = r.layer
        att, ff = layer.attention, layer.feed_forward
        ds2 = _ln_bwd(grads, layer.final_layer_norm, dx, r.s2, r.lnB)
        dt2 = ds2
        if r.d_ffo[0] > 0:
            dt2 = L.empty(ds2.shape, act16(), ds2)
            H.dropout_bf16(ds2, dt2, *r.d_ffo)
        _lin_grads(grads, ff.output_dense, r.h, dt2, M, 768, 3072, deferred, "ffn2")
        dh = L.linear_dgrad(dt2, M, r.ff2_wt, 3072)
        du = L.empty(dh.shape, act16(), dh)
        if r.d_int[0] > 0:
            H.gelu_bwd_dropout(dh, r.u, du, *r.d_int)      # dropout mask and GELU derivative in one pass
        else:
            H.gelu_bwd(dh, r.u, du)
        _lin_grads(grads, ff.intermediate_dense, r.xa, du, M, 3072, 768, deferred, "ffn1")
        dxa = L.linear_dgrad(du, M, r.ff1_wt, 768, residual=ds2)
        ds1 = _ln_bwd(grads, layer.layer_norm, dxa, r.s1, r.lnA)
        dt1 = ds1
        if r.d_out[0] > 0:
            dt1 = L.empty(ds1.shape, act16(), ds1)
            H.dropout_bf16(ds1, dt1, *r.d_out)
        _lin_grads(grads, att.out_proj, r.ctx, dt1, M, 768, 768, deferred, "out")
        dctx = L.linear_dgrad(dt1, M, r.out_wt, 768)
        dqkv = _attention_bwd(dctx, r.qkv, r.ctx, r.P, B, T, Tp, att.scaling, r.d_att)
        # q/k/v projections share one fused weight gradient
        qkv_lins = (att.q_proj, att.k_proj, att.v_proj)
        need = any(p.requires_grad for p in (att.q_proj.weight, att.k_proj.weight, att.v_proj.weight))
        if need and deferred is not None and all(l.weight.requires_grad and l.bias.requires_grad for l in qkv_lins):
            deferred.add("qkv", qkv_lins, r.x_in, dqkv, 2304, 768)
        elif need:
            dw, db = L.linear_wgrad(r.x_in, dqkv, M, 2304, 768)
            for i, lin in enumerate((att.q_proj, att.k_proj, att.v_proj)):
                if lin.weight.requires_grad:
                    grads[lin.weight] = dw[i * 768:(i + 1) * 768]
                    grads[lin.bias] = db[i * 768:(i + 1) * 768]
        dx = L.linear_dgrad(dqkv, M, r.qkv_wt, 768, residual=ds1)
    if deferred is not None:
        deferred.flush(grads)
    if t.d_tr[0] > 0:
        H.dropout_bf16(dx, dx, *t.d_tr)
    dx1 = _ln_bwd(grads, tr.layer_norm, dx, t.x1, t.ln1)
    # x1 = x0 + gelu(posconv(x0) + b)
    pc = tr.pos_conv_embed
    conv = pc.conv
    geom = t.pc_geom
    Cig = 768 // pc.groups
    du = L.empty((M, 768), act16(), dx1)
    H.gelu_bwd(dx1, t.pc_pre, du)
    if conv.weight_v.requires_grad:
        gw = L.conv_wgrad_raw(t.x0, du, geom)                      # [768][K][Cig]
        dv, dg, ws = L.empty(conv.weight_v.shape, f32, du), L.empty((pc.kernel,), f32, du), L.empty((pc.kernel,), f32, du)
        H.weightnorm_bwd(gw, conv.weight_v, conv.weight_g.reshape(-1), t.wn_norm, 768, Cig, pc.kernel, dv, dg, ws)
        grads[conv.weight_v], grads[conv.weight_g] = dv, dg.view(1, 1, -1)
        db = L.empty((768,), f32, du)
        H.colsum_bf16(du, M, 768, 768, db)
        grads[conv.bias] = db
    # dgrad of the grouped conv, + dx1 (residual path)
    wd = L.empty((pc.groups, Cig, pc.kernel, geom.Cog), act16(), du)
    wfp4 = t.wfp.view(pc.groups, geom.Cog, pc.kernel, Cig)
    for gi in range(pc.groups):
        # [Cog][K][Cig] -> [Cig][K][Cog]: transpose of a [Cog] x [K*Cig] matrix regrouped per tap
        H.transpose_bf16(wfp4[gi], Cig, pc.kernel * Cig, wd[gi], geom.Cog, pc.kernel * geom.Cog, pc.kernel, geom.Cog, Cig,
                         r_pad=geom.Cog)
    dx0 = L.empty((M, 768), act16(), du)
    H.igemm(du, wd, dx0, M, Cig, geom.Kd, geom.g_dgrad(), geom.Kd, 768, b_rows=Cig, residual=dx1, ldr=768,
            nbatch=pc.groups, inner=1, a_s=(geom.Cog, 0), b_s=(Cig * geom.Kd, 0), c_s=(Cig, 0))
    if t.d_fp[0] > 0:
        H.dropout_bf16(dx0, dx0, *t.d_fp)
    _lin_grads(grads, fp.projection, t.xln, dx0, M, 768, 512)
    dxln = L.linear_dgrad(dx0, M, t.proj_wt, 512)
    return _ln_bwd(grads, fp.layer_norm, dxln, t.feat, t.ln0)


class Wav2Vec2Fn(torch.autograd.Function):
    """wave fp32 (B,L) -> (B,T,num_out) fp32 [full] or (B,T,512) fp32 [features only]."""

    @staticmethod
    def forward(ctx, wave, model, full, want_grad, *params):
        if not wave.is_cuda:
            raise H.PeppaHipError("peppa_amd.audio needs a CUDA/HIP tensor (no CPU fallback)")
        wave = wave.contiguous().float()
        save = want_grad and any(ctx.needs_input_grad)  # grad mode is always off inside Function.forward
        fe = model.feature_extractor
        B = wave.shape[0]
        with torch.no_grad():
            feat, T, fe_tape = _fe_forward(fe, wave, save and any(p.requires_grad for p in fe.parameters()))
            if full:
                out, enc_tape = _enc_forward(model.encoder, feat, B, T, save, training=model.training)
                result = out.view(B, T, -1)
            else:
                enc_tape = None
                res = L.empty(feat.shape, f32, feat)
                H.cast_bf16_to_f32(feat, res)
                result = res.view(B, T, 512)
        ctx.model, ctx.full, ctx.fe_tape, ctx.enc_tape, ctx.params = model, full, fe_tape, enc_tape, params
        ctx.precision = H.precision()     # the backward pass runs on the library that produced the tape
        return result

    @staticmethod
    def backward(ctx, dout):
        model, grads = ctx.model, grad_dict()
        H.set_precision(ctx.precision)
        fe = model.feature_extractor
        dout = dout.contiguous().float()
        with torch.no_grad(), L.ZeroPool("audio", dout.device):
            if ctx.full:
                t = ctx.enc_tape
                dfeat = _enc_backward(model.encoder, t, dout.view(t.M, -1), grads)
            else:
                dfeat = L.empty((dout.numel() // 512, 512), act16(), dout)
                H.cast_f32_to_bf16(dout.view(-1, 512), dfeat)
            if any(p.requires_grad for p in fe.parameters()):
                _fe_backward(fe, ctx.fe_tape, dfeat, grads)
        return (None, None, None, None) + tuple(grads.get(p) for p in ctx.params)
