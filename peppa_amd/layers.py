"""Forward/backward primitives of the hot path, expressed over the C ABI (peppa_amd.hip).

Each primitive is an explicit (forward, backward) pair on channels-last 16-bit activations;
the encoders (video.py / audio.py) chain them inside one autograd.Function per encoder, so
activation buffers, weight operand layouts and streams are managed by hand instead of by a
tracing compiler.  No torch arithmetic happens here: torch only allocates.
"""
import weakref

import torch

from . import hip as H
from .hip import rup, act16, f32

CP = 16  # channel padding of activation tensors


def cpad(c):
    return rup(c, CP)


def empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


class ZeroPool:
    """One pre-zeroed fp32 arena per backward pass instead of hundreds of tiny fill kernels: the wgrad /
    reduction outputs that must start at zero are carved out of it.  The size is learned on the first pass
    (which falls back to torch.zeros) and the arena is a fresh allocation every pass, because the gradients
    handed to autograd are views into it and stay alive until the next zero_grad."""
    _sizes = {}
    _active = []

    def __init__(self, key, device):
        self.key, self.device = (key, str(device)), device
        self.buf, self.off, self.need = None, 0, 0

    def __enter__(self):
        n = ZeroPool._sizes.get(self.key, 0)
        if n:
            self.buf = torch.zeros(n, dtype=f32, device=self.device)
        ZeroPool._active.append(self)
        return self

    def __exit__(self, *exc):
        ZeroPool._active.pop()
        ZeroPool._sizes[self.key] = self.need
        return False

    def take(self, numel):
        n = (numel + 63) // 64 * 64          # keep every slice 256-byte aligned
        self.need += n
        if self.buf is None or self.off + n > self.buf.numel():
            return None
        out = self.buf[self.off:self.off + numel]
        self.off += n
        return out


def zeros(shape, dtype, like):
    if dtype == f32 and ZeroPool._active and ZeroPool._active[-1].device == like.device:
        numel = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            numel *= d
        t = ZeroPool._active[-1].take(numel)
        if t is not None:
            return t.view(shape)
    return torch.zeros(shape, dtype=dtype, device=like.device)


class ConvGeom:
    """Geometry of one convolution on a channels-last tensor [B][Ti][Hi][Wi][cstride]."""

    def __init__(self, B, in_thw, Ci, Co, k, s, p, groups=1, in_cstride=None, out_cstride=None, cg_in=None, To=None, Wo=None):
        self.B, self.Ci, self.Co, self.k, self.s, self.p, self.groups = B, Ci, Co, tuple(k), tuple(s), tuple(p), groups
        self.Ti, self.Hi, self.Wi = in_thw
        self.To = To if To is not None else (self.Ti + 2 * p[0] - k[0]) // s[0] + 1  # To override: pos-conv drops its last frame
        self.Ho = (self.Hi + 2 * p[1] - k[1]) // s[1] + 1
        self.Wo = Wo if Wo is not None else (self.Wi + 2 * p[2] - k[2]) // s[2] + 1  # Wo override: paired-pixel stem
        self.pairs = None     # (Ci, kw, pw) of the ORIGINAL convolution when this is its paired-pixel form (ConvGeom.paired_stem)
        self.taps = k[0] * k[1] * k[2]
        self.Cig, self.Cog = Ci // groups, Co // groups
        # channels read per tap / written per group
        self.cg_in = cg_in if cg_in is not None else (cpad(Ci) if groups == 1 else self.Cig)
        self.cg_out = cpad(Co) if groups == 1 else self.Cog
        self.in_cstride = in_cstride if in_cstride is not None else cpad(Ci)
        self.out_cstride = out_cstride if out_cstride is not None else cpad(Co)
        self.M = B * self.To * self.Ho * self.Wo
        self.Min = B * self.Ti * self.Hi * self.Wi
        self.Kf = self.taps * self.cg_in     # forward reduce length
        self.Kd = self.taps * self.cg_out    # dgrad reduce length
        self.nblk = (self.M + 127) // 128

    @property
    def out_thw(self):
        return (self.To, self.Ho, self.Wo)

    @staticmethod
    def paired_stem(B, in_thw, Ci, Co, k, s, p):
        """A first convolution with <= 4 input channels and stride 2 along W, over an input stored with FOUR channels per
        pixel: the pixel pairs (2 j, 2 j + 1) are 8-channel chunks [W/2][8] of the same memory, and the convolution is the
        stride-1 one over pairs with kwp = 4 (for kw = 7, pw = 3) taps per kernel row (include/peppa_hip.h,
        pp_prep_conv_weight_pairs).  An 8-channel chunk then carries two real pixels instead of one and five zeros: the
        reduce length of the (1,7,7) stem drops from 49 x 8 to 28 x 8.  None if the shape does not pair."""
        T, Hh, W = in_thw
        if Ci > 4 or s[2] != 2 or W % 2 != 0:
            return None
        kw, pw = k[2], p[2]
        lo, hi = (-pw) // 2, (kw - 1 - pw) // 2            # (floor division)
        kwp, pj = hi - lo + 1, -lo
        wo = (W + 2 * pw - kw) // 2 + 1
        g = ConvGeom(B, (T, Hh, W // 2), 8, Co, (k[0], k[1], kwp), (s[0], s[1], 1), (p[0], p[1], pj), in_cstride=8, cg_in=8, Wo=wo)
        g.pairs = (Ci, kw, pw)
        return g

    def g_fwd(self):
        return H.gather_conv(H.CONV_FWD, (self.To, self.Ho, self.Wo), (self.Ti, self.Hi, self.Wi), self.k, self.s,
                             self.p, self.cg_in, self.in_cstride)

    def g_dgrad(self):
        return H.gather_conv(H.CONV_DGRAD, (self.Ti, self.Hi, self.Wi), (self.To, self.Ho, self.Wo), self.k, self.s,
                             self.p, self.cg_out, self.out_cstride)


# ---- operand cache ------------------------------------------------------------------------------------------------
# The 16-bit operand layouts of a weight only change when the weight does.  With gradient accumulation (the reference
# trains with accumulate_grad_batches: 8, hparams_base.yaml:42), in validation and in any forward-only loop the same
# masters are re-laid for every micro-batch (~1.4 ms of small launches per pass): keep them until the masters move.
# A master moves (i) through torch (`load_state_dict`, `copy_`, ...: its `_version` changes) or (ii) through
# pp_bertadam_step, which writes through raw pointers: BertAdam.step() bumps WEIGHT_EPOCH.
# Off by default: a caller that edits `p.data` in place changes a master without either signal.  peppa_amd.trainer.Trainer
# owns the loop (weights move only in optimizer.step() and checkpoint loads) and switches it on.
WEIGHT_EPOCH = 0
_OPERANDS = {}
_OPERANDS_EPOCH = [0]
CACHE_OPERANDS = False
PREP_PLAN = True         # one launch for a tower's convolution operands (PrepPlan); False: two launches per convolution


def weights_changed():
    """Called by whatever rewrites parameters behind torch's back (BertAdam.step)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def _operand_slot(key, params):
    return (key, H.precision()) + tuple(p.data_ptr() for p in params), tuple(p._version for p in params)


def cached_operands(key, params, build):
    """build() -> operands of `params` (weights), memoised until one of them changes; key names the layout.  One entry per
    (layout, weight): a new version of the weight replaces it."""
    if not CACHE_OPERANDS:
        return build()
    if _OPERANDS_EPOCH[0] != WEIGHT_EPOCH:
        _OPERANDS.clear()
        _OPERANDS_EPOCH[0] = WEIGHT_EPOCH
    slot, versions = _operand_slot(key, params)
    hit = _OPERANDS.get(slot)
    if hit is None or hit[0] != versions:
        if len(_OPERANDS) > 4096:
            _OPERANDS.clear()
        hit = _OPERANDS[slot] = (versions, build())
    return hit[1]


def _conv_operand_key(geom, need_dgrad):
    return ("conv", geom.Co, geom.Cig, geom.taps, geom.cg_in, geom.cg_out, geom.groups, geom.Ci, bool(need_dgrad))


class PrepPlan:
    """The convolution weight operands of a whole tower in ONE launch (pp_prep_conv_weight_multi) instead of two 4-8 us
    launches per convolution on the tower's critical stream.  `with PrepPlan(key):` around a tower's forward pass: the first
    pass records which (weight, geometry) pairs prep_conv_weights is asked for; later passes build all of them on entry
    (those the operand cache does not already hold) and hand them out as the layers ask.  A recorded pair that is not asked
    for again costs one wasted conversion; a pair that was not recorded is built on its own as before."""
    _plans = {}
    _active = []

    def __init__(self, key):
        self.key, self.seen, self.ready, self._keep = key, [], {}, None

    def __enter__(self):
        if PREP_PLAN:
            self._build(PrepPlan._plans.get(self.key, ()))
        PrepPlan._active.append(self)
        return self

    def __exit__(self, *exc):
        PrepPlan._active.pop()
        if exc[0] is None:
            PrepPlan._plans[self.key] = self.seen
        self.ready = {}
        return False

    def _build(self, entries):
        jobs = []
        for wref, geom, need_dgrad in entries:
            w = wref()
            if w is None or not w.is_cuda or not w.is_contiguous():
                continue
            key = _conv_operand_key(geom, need_dgrad)
            if (w.data_ptr(), key) in self.ready:
                continue
            if CACHE_OPERANDS and _OPERANDS_EPOCH[0] == WEIGHT_EPOCH:
                slot, versions = _operand_slot(key, (w,))
                hit = _OPERANDS.get(slot)
                if hit is not None and hit[0] == versions:
                    continue
            wf = empty((geom.Co, geom.taps, geom.cg_in), act16(), w)
            jobs.append((w, wf, geom.Co, geom.Cig, geom.taps, geom.Co, geom.cg_in, False))
            wd = None
            if need_dgrad:
                wd = empty((geom.Ci, geom.taps, geom.cg_out), act16(), w)
                jobs.append((w, wd, geom.Co, geom.Ci, geom.taps, geom.Ci, geom.cg_out, True))
            self.ready[(w.data_ptr(), key)] = (wf, wd)
        if jobs:
            self._keep = H.prep_conv_weight_multi(jobs, jobs[0][0].device)

    def take(self, w, geom, need_dgrad):
        self.seen.append((weakref.ref(w), geom, need_dgrad))
        return self.ready.pop((w.data_ptr(), _conv_operand_key(geom, need_dgrad)), None)


def prep_conv_weights(w, geom, need_dgrad=True):
    """fp32 master [Co][Ci/groups][taps...] -> 16-bit operands (forward, dgrad)."""
    if geom.pairs is not None:
        assert not need_dgrad, "a paired-pixel stem is a first layer: no data gradient"
        key = ("conv-pairs", geom.Co, geom.k, geom.pairs)
        return cached_operands(key, (w,), lambda: (_prep_conv_weights_pairs(w, geom), None))
    if geom.groups == 1 and PrepPlan._active:
        ready = PrepPlan._active[-1].take(w, geom, need_dgrad)
        if ready is not None:
            return cached_operands(_conv_operand_key(geom, need_dgrad), (w,), lambda: ready)
    return cached_operands(_conv_operand_key(geom, need_dgrad), (w,), lambda: _prep_conv_weights(w, geom, need_dgrad))


def _prep_conv_weights_pairs(w, geom):
    Ci, kw, pw = geom.pairs
    wf = empty((geom.Co, geom.taps, 8), act16(), w)
    H.prep_conv_weight_pairs(w, wf, geom.Co, Ci, geom.k[0] * geom.k[1], kw, pw)
    return wf


def _prep_conv_weights(w, geom, need_dgrad=True):
    Co, Cig, taps = geom.Co, geom.Cig, geom.taps
    wf = empty((Co, taps, geom.cg_in), act16(), w)
    H.prep_conv_weight(w, wf, Co, Cig, taps, Co, geom.cg_in)
    wd = None
    if need_dgrad:
        if geom.groups == 1:
            wd = empty((geom.Ci, taps, geom.cg_out), act16(), w)
            H.prep_conv_weight(w, wd, Co, geom.Ci, taps, geom.Ci, geom.cg_out, transpose_io=True)
        else:
            # per group g: rows = ci (Cig), reduce = (tap, co in group)
            wd = empty((geom.groups, Cig, taps, geom.Cog), act16(), w)
            for gi in range(geom.groups):
                H.prep_conv_weight(w[gi * geom.Cog:(gi + 1) * geom.Cog], wd[gi], geom.Cog, Cig, taps, Cig, geom.Cog,
                                   transpose_io=True)
    return wf, wd


def can_fuse_bn_apply(geom):
    """May the BatchNorm unit that FEEDS this convolution skip its apply pass -- i.e. do both the forward kernel and the
    weight-gradient kernel this geometry dispatches to apply z = relu?(y * scale + shift) themselves, on their LDS windows
    (pp_igemm_desc.a_bn_*, pp_wgrad_desc.x_bn_*)?  True for the layer-1 temporal convolutions of r2plus1d at the C2 shapes."""
    if geom.groups != 1:
        return False
    return (H.igemm_abn_supported(geom.M, geom.out_cstride, geom.Kf, geom.g_fwd()) and
            H.wgrad_xbn_supported(geom.M, geom.Co, geom.Kf, geom.g_fwd(), geom.out_cstride))


STEM_WINDOW = True        # A/B switch (tools/ab_step.py stem_window): pp_stem_pairs_fwd for the paired-pixel stem


def _stem_window_ok(geom):
    """The (1,7,7) stride-(1,2,2) padding-(0,3,3) stem over pixel pairs, frames up to 64 pairs (128 pixels) wide, <= 48 output
    channels (r2plus1d_18's 45): what pp_stem_pairs_fwd is built for."""
    return (geom.pairs is not None and geom.groups == 1 and geom.pairs[1] == 7 and geom.pairs[2] == 3 and geom.k == (1, 7, 4) and
            geom.s == (1, 2, 1) and geom.p[:2] == (0, 3) and geom.Wi <= 64 and geom.Wo == geom.Wi and geom.Co <= 48 and
            geom.in_cstride == 8 and geom.To == geom.Ti)


def conv_fwd(x, geom, wf, *, stats=False, bias=None, act=H.ACT_NONE, out=None, pre=None, x_bn=None):
    """y[M][out_cstride] = conv(x); optional per-column partial sums for BatchNorm.
    x_bn = (scale, shift, relu): x is the raw output of a BatchNorm unit whose apply pass was skipped (can_fuse_bn_apply)."""
    y = out if out is not None else empty((geom.M, geom.out_cstride), act16(), x)
    if STEM_WINDOW and _stem_window_ok(geom) and bias is None and act == H.ACT_NONE and pre is None and x_bn is None:
        # the paired-pixel stem as a window kernel: the input is read once instead of 28 x 16 bytes per output row
        images = geom.B * geom.Ti
        partials = empty((H.stem_pairs_stat_rows(images, geom.Hi), 2, geom.out_cstride), f32, x) if stats else None
        H.stem_pairs_fwd(x, wf, y, partials, images, geom.Hi, geom.Wi, geom.Co, geom.out_cstride, geom.out_cstride)
        return y, partials
    # (rows of partial statistics: one per 128 output rows, a few more where the temporal window kernel's tiles do not
    # divide the clip evenly -- callers pass partials.shape[0] on to bn_fwd, not geom.nblk)
    partials = None
    if stats:
        nrows = H.igemm_stat_rows(geom.M, geom.out_cstride, geom.Kf, geom.g_fwd(), bna=x_bn is not None) if geom.groups == 1 else geom.nblk
        partials = empty((nrows, 2, geom.out_cstride), f32, x)
    if geom.groups == 1:
        H.igemm(x, wf, y, geom.M, geom.out_cstride, geom.Kf, geom.g_fwd(), geom.Kf, geom.out_cstride,
                b_rows=geom.Co, bias=bias, act=act, Cpre=pre, colstats=partials, ldstat=geom.out_cstride, bna=x_bn)
    else:
        G = geom.groups
        H.igemm(x, wf, y, geom.M, geom.Cog, geom.Kf, geom.g_fwd(), geom.Kf, geom.out_cstride, b_rows=geom.Cog,
                bias=bias, act=act, Cpre=pre, nbatch=G, inner=1, a_s=(geom.Cig, 0), b_s=(geom.Cog * geom.Kf, 0),
                c_s=(geom.Cog, 0), bias_s=(geom.Cog, 0))
    return y, partials


def _parity_classes(geom):
    """Stride-2 data gradient = one dense unit-stride problem per output-parity class: positions r = s*r' + q
    only see taps d = d0 + s*i with d0 = (q + p) mod s, and read dY at r' + (q + p - d0)/s - i."""
    import itertools
    dims = list(zip((geom.Ti, geom.Hi, geom.Wi), geom.k, geom.s, geom.p))
    per_dim = []
    for R, k, s, p in dims:
        opts = []
        for q in range(s):
            d0 = (q + p) % s
            taps = list(range(d0, k, s))
            Rc = (R - q + s - 1) // s if R > q else 0
            opts.append((q, taps, (q + p - d0) // s, Rc))
        per_dim.append(opts)
    return list(itertools.product(*per_dim))


def _conv_dgrad_strided(dy, geom, wd, residual):
    classes = _parity_classes(geom)
    empty_class = any(len(t) == 0 for cls in classes for (_, t, _, _) in cls)
    assert not (empty_class and residual is not None)
    dx = (zeros if empty_class else empty)((geom.Min, geom.in_cstride), act16(), dy)
    kt, kh, kw = geom.k
    for cls in classes:
        (qt, tt, ct, Rt), (qh, th, ch, Rh), (qw, tw, cw, Rw) = cls
        if not (tt and th and tw) or Rt * Rh * Rw == 0:
            continue
        sel = [(a * kh + b) * kw + c for a in tt for b in th for c in tw]
        wsel = empty((geom.Ci, len(sel), geom.cg_out), act16(), dy)
        H.select_taps(wd, wsel, geom.Ci, geom.taps, geom.cg_out, sel)
        g = H.gather_conv(H.CONV_DGRAD, (Rt, Rh, Rw), (geom.To, geom.Ho, geom.Wo), (len(tt), len(th), len(tw)), (1, 1, 1),
                          (ct, ch, cw), geom.cg_out, geom.out_cstride)
        K = len(sel) * geom.cg_out
        H.igemm(dy, wsel, dx, geom.B * Rt * Rh * Rw, geom.in_cstride, K, g, K, geom.in_cstride, b_rows=geom.Ci,
                residual=residual, ldr=geom.in_cstride,
                omap=((geom.Ti, geom.Hi, geom.Wi), geom.s, (qt, qh, qw)))
    return dx


WIN_S2D = True    # stride-(1,2,2) spatial data gradients as ONE window-kernel launch that reads dy once (pp_set_option "win_s2d")


def _s2d_window(geom):
    """Does pp_igemm take this data gradient with the window kernel's S2D form (csrc/igemm_win.hip: a (1,3,3) convolution
    with stride (1,2,2), pad (0,1,1), output at most 63 wide)?  Then conv_dgrad issues ONE launch over the whole gather
    instead of one gather-kernel launch per output parity class."""
    return (WIN_S2D and geom.k == (1, 3, 3) and geom.s == (1, 2, 2) and geom.p == (0, 1, 1) and geom.Wo + 1 <= 64 and
            geom.Min >= H.WIN_IGEMM_DEFAULT and geom.Ho == (geom.Hi + 1) // 2 and geom.Wo == (geom.Wi + 1) // 2)


MASKED_STRIDED_DGRAD = False   # A/B: stride-2 data gradients as ONE launch with masked taps instead of parity classes
FUSE_BN_BWD_REDUCE = False   # consumer BatchNorm-backward sums in the data-gradient epilogue: built, tested, and OFF -- measured slower


def conv_dgrad(dy, geom, wd, *, residual=None, consumer=None):
    """dx[Min][in_cstride] = conv^T(dy) (+ residual).

    consumer = (y, z or None, BNSaved, relu) of the BatchNorm unit that receives dx as its dz: its backward sums
    (sum g, sum g * xhat) are then accumulated in the epilogue of the window kernels while the tile is on chip, and
    `consumer_partials(dx)` hands them to bn_bwd, which skips its own pass over dz."""
    if geom.groups == 1 and max(geom.s) == 2 and not MASKED_STRIDED_DGRAD and not _s2d_window(geom):
        return _conv_dgrad_strided(dy, geom, wd, residual)
    dx = empty((geom.Min, geom.in_cstride), act16(), dy)
    if geom.groups == 1:
        bnr = None
        if consumer is not None and FUSE_BN_BWD_REDUCE and geom.Min >= 1024:
            y, z, sv, relu = consumer
            nblk = (geom.Min + 255) // 256
            partials = empty((nblk, 2, geom.in_cstride), f32, dy)
            bnr = (y, z, sv.mean, sv.rstd, sv.scale, sv.shift, relu, partials)
        fused = H.igemm(dy, wd, dx, geom.Min, geom.in_cstride, geom.Kd, geom.g_dgrad(), geom.Kd, geom.in_cstride,
                        b_rows=geom.Ci, residual=residual, ldr=geom.in_cstride, bnr=bnr)
        if fused:
            dx._bnr = (bnr[-1], nblk)     # rides on the tensor object to the bn_bwd call that receives dx as its dz
    else:
        assert residual is None
        G = geom.groups
        H.igemm(dy, wd, dx, geom.Min, geom.Cig, geom.Kd, geom.g_dgrad(), geom.Kd, geom.in_cstride, b_rows=geom.Cig,
                nbatch=G, inner=1, a_s=(geom.Cog, 0), b_s=(geom.Cig * geom.Kd, 0), c_s=(geom.Cig, 0))
    return dx


def conv_wgrad_raw(x, dy, geom, x_bn=None):
    """fp32 gradient in operand layout [Co][taps][cg_in]."""
    gw = zeros((geom.Co, geom.taps, geom.cg_in), f32, x)
    if STEM_WINDOW and x_bn is None and not H.DETERMINISTIC and _stem_window_ok(geom):
        # the paired-pixel stem on its window kernel (fp32 atomics across workgroups: the deterministic mode keeps pp_wgrad's slabs)
        H.stem_pairs_wgrad(x, dy, gw, geom.B * geom.Ti, geom.Hi, geom.Wi, geom.Co, geom.out_cstride)
        return gw
    if geom.groups == 1:
        H.wgrad(x, dy, gw, geom.M, geom.Co, geom.Kf, geom.g_fwd(), geom.out_cstride, geom.Kf, x_bn=x_bn)
    else:
        H.wgrad(x, dy, gw, geom.M, geom.Cog, geom.Kf, geom.g_fwd(), geom.out_cstride, geom.Kf, nbatch=geom.groups,
                x_s=geom.Cig, dy_s=geom.Cog, dw_s=geom.Cog * geom.Kf)
    return gw


def conv_wgrad(x, dy, geom, w_shape, x_bn=None):
    gw = conv_wgrad_raw(x, dy, geom, x_bn)
    dw = empty(w_shape, f32, x)
    if geom.pairs is not None:
        Ci, kw, pw = geom.pairs
        H.unprep_conv_grad_pairs(gw, dw, geom.Co, Ci, geom.k[0] * geom.k[1], kw, pw)
    else:
        H.unprep_conv_grad(gw, dw, geom.Co, geom.Cig, geom.taps, geom.cg_in)
    return dw


# ---- BatchNorm (train mode) -------------------------------------------------------------------------
class BNSaved:
    __slots__ = ("mean", "rstd", "scale", "shift", "count", "C", "Cp")


# SyncBN (off by default like the reference's DDP run): batch statistics over ALL ranks.  SYNC_BN_REDUCE is the
# collective -- `torch.distributed.all_reduce(t)` (RCCL, SUM, in place) -- and SYNC_BN_WORLD the number of ranks;
# peppa_amd.dist.enable_sync_bn() sets both (tests plug in a stand-in to play two ranks in one process).
SYNC_BN_REDUCE = None
SYNC_BN_WORLD = 1
# Deterministic mode (hip.set_deterministic): the ranks exchange their per-block PARTIAL rows instead of their sums
# (`all_gather_into_tensor(out, t)`), and every rank finalizes over the concatenation -- exactly the partial rows, in
# exactly the order, that ONE process running the global batch reduces (blocks are fixed-size runs of rows and the batch
# is rank-major), so the statistics, and with them every activation and data gradient, agree bit for bit with that
# process.  Costs world x the traffic of the sums (29 MB per rank for a layer-1 unit at batch 64): a checking mode.
SYNC_BN_GATHER = None


def _global_sums(partials, nblk, Cp):
    """[nblk][2][Cp] per-rank partial sums -> [1][2][Cp] sums over every rank's rows (one small all-reduce)."""
    tot = empty((1, 2, Cp), f32, partials)
    H.partials_sum(partials, nblk, Cp, empty((64, 2, Cp), f32, partials), tot)
    SYNC_BN_REDUCE(tot)
    return tot


def _global_partials(partials, nblk, Cp):
    """[nblk][2][Cp] partial rows of this rank -> ([world * nblk][2][Cp] of every rank, rank-major, world * nblk)."""
    out = empty((SYNC_BN_WORLD * nblk, 2, Cp), f32, partials)
    SYNC_BN_GATHER(out, partials[:nblk].contiguous())
    return out, SYNC_BN_WORLD * nblk


def _ordered_sync():
    return H.DETERMINISTIC and SYNC_BN_GATHER is not None


def bn_fwd(y, partials, nblk, count, bn, *, relu, residual=None, eps=1e-5, momentum=0.1, update_running=True, apply=True):
    """z = relu?(bn(y) (+residual)); `bn` has .weight .bias .running_mean .running_var.
    apply=False: statistics / scale / shift only, z = None -- the consumer applies them itself (can_fuse_bn_apply)."""
    C = bn.weight.numel()
    Cp = y.shape[1]
    sv = BNSaved()
    sv.count, sv.C, sv.Cp = count, C, Cp
    sv.mean, sv.rstd, sv.scale, sv.shift = (empty((Cp,), f32, y) for _ in range(4))
    if partials is None:  # eval mode: running statistics (no backward through this path)
        H.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, eps, C, Cp, sv.scale, sv.shift)
    else:
        if SYNC_BN_REDUCE is not None:      # statistics of the global batch: sums and count over all ranks
            if _ordered_sync():
                partials, nblk = _global_partials(partials, nblk, Cp)
                count = count * SYNC_BN_WORLD
            else:
                partials, nblk, count = _global_sums(partials, nblk, Cp), 1, count * SYNC_BN_WORLD
            sv.count = count
        ws = empty((64, 2, Cp), f32, y) if nblk > 256 else None
        H.bn_finalize(partials, nblk, Cp, count, C, Cp, bn.weight, bn.bias, eps, momentum,
                      bn.running_mean if update_running else None, bn.running_var if update_running else None,
                      sv.mean, sv.rstd, sv.scale, sv.shift, ws)
    if not apply:
        assert residual is None
        return None, sv
    z = empty(y.shape, act16(), y)
    H.bn_apply(y, sv.scale, sv.shift, residual, relu, z, y.shape[0], Cp)
    return z, sv


def bn_bwd(dz, y, z, sv, gamma, *, relu, want_dres=False):
    """returns dy, dres (masked dz, for the skip connection), dgamma, dbeta.
    z=None (units without a residual input): the ReLU mask is recomputed from y, saving one stream."""
    M, Cp = y.shape
    ready = getattr(dz, "_bnr", None)
    if ready is not None:
        partials, nblk = ready             # (sum g, sum g * xhat) per 256 rows, from the data-gradient epilogue
    else:
        # (deterministic mode: fixed 64-row blocks whatever M is, so that a rank's blocks are blocks of the global batch)
        nblk = (M + 63) // 64 if H.DETERMINISTIC else min(2048, (M + 63) // 64)
        partials = empty((nblk, 2, Cp), f32, y)
        H.bn_bwd_reduce(dz, y, z, sv.mean, sv.rstd, sv.scale, sv.shift, relu, partials, nblk, M, Cp)
    dgamma, dbeta = empty((sv.C,), f32, y), empty((sv.C,), f32, y)
    coef = empty((3, Cp), f32, y)
    # (more than 256 partial rows: two levels -- 64 slices in parallel, then 64 rows; 9 blocks walking 2048 rows took 35 us)
    # (not in deterministic mode: the first level sums in fp32 over slices of THIS rank's rows, and the mode promises the same
    # bits for one process on the global batch and for any number of ranks -- there the rows are walked in one double-precision pass)
    ws = empty((64 * 2 * Cp,), f32, y) if nblk > 256 and not H.DETERMINISTIC else None
    H.bn_bwd_finalize(partials, nblk, sv.count, sv.C, Cp, gamma, sv.rstd, dgamma, dbeta, coef, ws)
    if SYNC_BN_REDUCE is not None:
        # dgamma / dbeta stay this rank's sums (the data-parallel all-reduce adds the ranks up); the coefficients of dy
        # -- the means of g and g * xhat -- are those of the global batch (sv.count already is the global count)
        scratch = empty((2, sv.C), f32, y)
        if _ordered_sync():
            gp, gn = _global_partials(partials, nblk, Cp)
            H.bn_bwd_finalize(gp, gn, sv.count, sv.C, Cp, gamma, sv.rstd, scratch[0], scratch[1], coef)
        else:
            H.bn_bwd_finalize(_global_sums(partials, nblk, Cp), 1, sv.count, sv.C, Cp, gamma, sv.rstd, scratch[0], scratch[1], coef)
    dy = empty(y.shape, act16(), y)
    dres = empty(y.shape, act16(), y) if want_dres else None
    H.bn_bwd_apply(dz, y, z, sv.mean, sv.rstd, coef, sv.scale, sv.shift, relu, dy, dres, M, Cp)
    return dy, dres, dgamma, dbeta


# ---- Linear (dense GEMM) ------------------------------------------------------------------------------
def prep_linear(w, need_dgrad=True):
    """w fp32 [N][K] -> (16-bit [N][Kp], 16-bit transposed [K][Np]) with Kp, Np multiples of 16."""
    N, K = w.shape
    Kp, Np = cpad(K), cpad(N)
    wf = empty((N, Kp), act16(), w)
    H.cast_pad_2d(w, wf, N, K, K, N, Kp)
    wt = None
    if need_dgrad:
        wt = empty((K, Np), act16(), w)
        H.cast_pad_2d(w, wt, K, N, K, K, Np, transpose=True)
    return wf, wt


class CastBatch:
    """Weight-operand preparation for many Linear layers in ONE launch (pp_cast_pad_2d_multi)."""

    def __init__(self):
        self.jobs = []
        self._keep = None

    def cast(self, w, out, rows, cols, ld_in, rows_out, cols_out, ld_out, transpose=False, out_f32=False):
        self.jobs.append((w, out, rows, cols, ld_in, rows_out, cols_out, ld_out, transpose, out_f32))

    def linear(self, w, need_dgrad=True):
        """Same operands as prep_linear: (16-bit [N][Kp], 16-bit transposed [K][Np] or None)."""
        N, K = w.shape
        Kp, Np = cpad(K), cpad(N)
        wf = empty((N, Kp), act16(), w)
        self.cast(w, wf, N, K, K, N, Kp, Kp)
        wt = None
        if need_dgrad:
            wt = empty((K, Np), act16(), w)
            self.cast(w, wt, K, N, K, K, Np, Np, transpose=True)
        return wf, wt

    def run(self, device):
        if self.jobs:
            self._keep = H.cast_pad_2d_multi(self.jobs, device)
        self.jobs = []


def linear_fwd(x, M, wf, N, *, bias=None, act=H.ACT_NONE, residual=None, pre=None, out=None, out_f32=False, dropout=None):
    """x 16-bit [M][Kp] -> y [M][Np] = dropout(act(x W^T + bias)) + residual; `pre` receives the pre-activation."""
    Kp = wf.shape[1]
    Np = cpad(N)
    y = out if out is not None else empty((M, Np), f32 if out_f32 else act16(), x)
    H.igemm(x, wf, y, M, N, Kp, H.gather_dense(x.shape[1]), Kp, Np, b_rows=N, bias=bias, act=act, residual=residual,
            ldr=Np, Cpre=pre, dropout=dropout)
    return y


def linear_dgrad(dy, M, wt, K, *, residual=None):
    """dx [M][Kp] = dy [M][Np] @ W ; wt is the transposed operand [K][Np]."""
    Np = wt.shape[1]
    Kp = cpad(K)
    dx = empty((M, Kp), act16(), dy)
    H.igemm(dy, wt, dx, M, K, Np, H.gather_dense(dy.shape[1]), Np, Kp, b_rows=K, residual=residual, ldr=Kp)
    return dx


def linear_wgrad(x, dy, M, N, K, *, want_bias=True):
    """dW fp32 [N][K], db fp32 [N] from x [M][Kp], dy [M][Np]."""
    Kp, Np = x.shape[1], dy.shape[1]
    # one zeroed arena for dW and db; the bias gradient is accumulated by the same wgrad launch
    arena = zeros((N * Kp + (N if want_bias else 0),), f32, x)
    gw = arena[:N * Kp].view(N, Kp)
    db = arena[N * Kp:] if want_bias else None
    H.wgrad(x, dy, gw, M, N, Kp, H.gather_dense(Kp), Np, Kp, dbias=db)
    if Kp != K:
        dw = empty((N, K), f32, x)
        H.copy_2d_f32(gw, Kp, dw, K, N, K)
    else:
        dw = gw
    return dw, db


_GROUP_TABLES = []     # (pinned host table, device table) of the last grouped launches: alive until their copies have run


def linear_wgrad_group(items, M, N, K, *, want_bias=True):
    """[(x [M][Kp], dy [M][Np])] x n -> [(dW fp32 [N][K], db fp32 [N] or None)] x n from ONE launch (pp_wgrad_desc.ptr_table):
    the same Linear shape of several layers.  Every (i, j) tile reduces its whole M, so nothing is summed across
    workgroups: no split atomics, bitwise reproducible, and 36 x n instead of 36 tiles for a 768 x 768 matrix."""
    n = len(items)
    if n == 0:
        return []
    x0, dy0 = items[0]
    Kp, Np = x0.shape[1], dy0.shape[1]
    per = N * Kp + (N if want_bias else 0)
    per_al = (per + 63) // 64 * 64
    arena = zeros((n * per_al,), f32, x0)
    outs, rows = [], []
    for i, (x, dy) in enumerate(items):
        if x.shape != x0.shape or dy.shape != dy0.shape or not (x.is_contiguous() and dy.is_contiguous()):
            raise H.PeppaHipError("linear_wgrad_group: every problem of a group has the same contiguous operand shapes")
        base = arena[i * per_al:(i + 1) * per_al]
        gw = base[:N * Kp].view(N, Kp)
        db = base[N * Kp:N * Kp + N] if want_bias else None
        rows.append([x.data_ptr(), dy.data_ptr(), gw.data_ptr(), db.data_ptr() if want_bias else 0])
        outs.append((gw, db))
    host = torch.tensor(rows, dtype=torch.int64).pin_memory()
    tab = host.to(x0.device, non_blocking=True)
    _GROUP_TABLES.append((host, tab))
    del _GROUP_TABLES[:-32]
    H.wgrad(x0, dy0, outs[0][0], M, N, Kp, H.gather_dense(Kp), Np, Kp, msplit=1, nbatch=n, dbias=outs[0][1], ptr_table=tab)
    if Kp != K:
        res = []
        for gw, db in outs:
            dw = empty((N, K), f32, x0)
            H.copy_2d_f32(gw, Kp, dw, K, N, K)
            res.append((dw, db))
        return res
    return outs


# ---- LayerNorm ---------------------------------------------------------------------------------------
def layernorm_fwd(x, ln, eps=1e-5):
    rows, D = x.shape
    y = empty(x.shape, act16(), x)
    mean, rstd = empty((rows,), f32, x), empty((rows,), f32, x)
    H.layernorm_fwd(x, ln.weight, ln.bias, eps, y, mean, rstd, rows, D)
    return y, (mean, rstd)


def layernorm_bwd(dy, x, ln, saved):
    rows, D = x.shape
    dx = empty(x.shape, act16(), x)
    dg, db = zeros((D,), f32, x), zeros((D,), f32, x)
    ws = empty((min(512, (rows + 15) // 16), 2, D), f32, x)   # per-workgroup partials (4 waves x 4 rows each)
    H.layernorm_bwd(dy, x, ln.weight, saved[0], saved[1], dx, dg, db, rows, D, ws)
    return dx, dg, db
