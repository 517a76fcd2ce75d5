"""Tensor-level wrappers over the C ABI (include/peppa_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function below
passes raw pointers + the current HIP stream to libpeppa_hip.so.  Nothing in this module
computes with torch ops, and there is no fallback when a tensor is not on the GPU.
"""
import ctypes as C
import torch

from . import _lib
from ._lib import call, Gather, IGemmDesc, WGradDesc, TensorList, PeppaHipError

DENSE, CONV_FWD, CONV_DGRAD = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
f32 = torch.float32
_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


def act16():
    """torch dtype of the 16-bit operands / activations of the library `call` currently dispatches to."""
    return _DTYPES[_lib.PRECISION]


def precision():
    return _lib.PRECISION


def set_precision(p):
    """"bf16" (libpeppa_hip.so, the default: BASELINE configs[1]) or "fp16" (libpeppa_hip_f16.so: the reference's
    `precision: 16`, BASELINE configs[4]); returns the previous setting."""
    p = {"16": "fp16", 16: "fp16", "half": "fp16", "float16": "fp16", "bfloat16": "bf16"}.get(p, p)
    if p not in _DTYPES:
        raise ValueError(f"precision must be 'bf16' or 'fp16', got {p!r}")
    prev, _lib.PRECISION = _lib.PRECISION, p
    return prev



def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise PeppaHipError("peppa_amd kernels need CUDA/HIP tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise PeppaHipError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def set_option(name, value):
    call("pp_set_option", name.encode(), int(value))


DETERMINISTIC = False


def set_deterministic(on=True):
    """pp_set_option("deterministic", ...) in every build of the library that is loaded or present + the workspaces the
    wrappers below then pass: every sum that crosses workgroups is taken in a fixed order instead of by fp32 atomics, so two
    runs of the same step give bitwise-identical losses, gradients and updates (include/peppa_hip.h; cost: DESIGN.md
    section 7).  Returns the previous setting.  `mi355x: {deterministic: true}` in the yaml / `run.py --deterministic`
    switch it on.  The mode is PROCESS-GLOBAL (a C option of the library, like torch.use_deterministic_algorithms): a
    model built with the yaml switch turns it on for every model of the process; callers that want it scoped restore
    the returned value in a `finally`.  A build that is not there (a bf16-only install) is skipped and gets the option
    when it is loaded later (`_lib.lib` applies `_lib.STICKY_OPTIONS`); the Python flag only changes after every C call
    has succeeded, so a failure cannot leave the wrappers and the library disagreeing about the workspaces."""
    global DETERMINISTIC
    import os
    on = bool(on)
    for prec in ("bf16", "fp16"):
        if prec in _lib._libs or os.path.exists(_lib.LIB_PATHS[prec]):
            h = _lib.lib(prec)
            if h.pp_set_option(b"deterministic", int(on)) != 0:
                raise PeppaHipError(f"pp_set_option(deterministic) [{prec}]: {h.pp_last_error().decode()}")
    _lib.STICKY_OPTIONS["deterministic"] = int(on)
    prev, DETERMINISTIC = DETERMINISTIC, on
    return prev


def rup(x, m):
    return (x + m - 1) // m * m


def gather_dense(lda):
    g = Gather()
    g.mode, g.lda = DENSE, lda
    g.Rt = g.Rh = g.Rw = g.Gt = g.Gh = g.Gw = g.kt = g.kh = g.kw = g.st = g.sh = g.sw = 1
    return g


def gather_conv(mode, R, G, k, s, p, cg, cstride):
    g = Gather()
    g.mode, g.lda = mode, 0
    g.Rt, g.Rh, g.Rw = R
    g.Gt, g.Gh, g.Gw = G
    g.kt, g.kh, g.kw = k
    g.st, g.sh, g.sw = s
    g.pt, g.ph, g.pw = p
    g.cg, g.cstride = cg, cstride
    return g


WIN_TALL_DEFAULT = 0       # pp_set_option("win_tall", n): 512-row window tiles for narrow-output data gradients (1: with enough rows, 2: always)
WGRAD_GROUP_RING_DEFAULT = 0   # pp_set_option("wgrad_group_ring", 1): grouped Linear weight gradients on the ring kernel's 128 x 256 tiles
RING_WGRAD_DEFAULT = 0     # pp_set_option("ring_wgrad", n): LDS-DMA ring weight gradient once M >= n rows (0 = never)
SW_WGRAD_DEFAULT = 4096   # pp_set_option("sw_wgrad", n): sliding-window wgrad of (1,3,3) stride-1 convs once M >= n (0 = never)
WIN_IGEMM_DEFAULT = 1024   # pp_set_option("win_igemm", n): window conv kernel for (1,3,3) stride-1 convs once M >= n (0 = never)
RING_IGEMM_DEFAULT = 128   # pp_set_option("ring_igemm", n): LDS-DMA ring GEMM once there are n 256-row tiles (0 = never)


# ---- optional in-process kernel timing (bench.py roofline): HIP events on the launch stream ----------
PROFILE_ON = False
PROFILE = {}
PROFILE_MIN_FLOP = 1.5e11   # only the big launches are timed: every timed launch costs two event records on its stream
_MODE_NAMES = {DENSE: "dense", CONV_FWD: "conv_fwd", CONV_DGRAD: "conv_dgrad"}


LAUNCH_LOG = None      # a list -> every pp_igemm / pp_wgrad call appends (entry, mode, M, N, K, nbatch, taps, strides, cg): ONE
                       # kernel launch each, in host order = rocprofv3's Dispatch_Id order (tools/prof_shapes.py joins them)
PROFILE_STREAM = None  # only launches on this stream are timed (the video trunk's stream in bench.py)
PROFILE_ONLY = None    # if set: only this kernel family is timed (bench.py picks it during warm-up)


def _profiled(key, flops, fn):
    if not PROFILE_ON or flops < PROFILE_MIN_FLOP or (PROFILE_ONLY is not None and key != PROFILE_ONLY):
        return fn()
    if PROFILE_STREAM is not None and torch.cuda.current_stream() != PROFILE_STREAM:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    PROFILE.setdefault(key, []).append((e0, e1, flops))


def profile_summary(key=None):
    """(flops, seconds, launches, name) of the kernel family with the largest total time (or of `key`), or None."""
    torch.cuda.synchronize()
    best = None
    for k, recs in PROFILE.items():
        if key is not None and k != key:
            continue
        secs = sum(a.elapsed_time(b) for a, b, _ in recs) * 1e-3
        fl = sum(f for _, _, f in recs)
        if best is None or secs > best[1]:
            best = (fl, secs, len(recs), k)
    return best


def _spatial_s1(g):
    return (g.mode != DENSE and (g.kt, g.kh, g.kw) == (1, 3, 3) and (g.st, g.sh, g.sw) == (1, 1, 1)
            and (g.Gt, g.Gh, g.Gw) == (g.Rt, g.Rh, g.Rw))


def _igemm_family(g):
    """Kernel family pp_igemm dispatches to with the default options (names as rocprofv3 prints them)."""
    return "igemm_win_kernel" if _spatial_s1(g) and (g.cg % 64 == 0 or g.cg % 48 == 0) else "igemm_kernel"


def _wgrad_family(g):
    if g.mode == CONV_FWD and _spatial_s1(g) and g.cg % 64 == 0:
        return "wgrad_sw_kernel"
    if g.mode == CONV_FWD and (g.kt, g.kh, g.kw) == (3, 1, 1) and (g.st, g.sh, g.sw) == (1, 1, 1) and g.cg >= 96:
        return "wgrad_tw_kernel"
    return "wgrad_kernel"


def igemm(A, Bt, Cout, M, N, K, g, ldb, ldc, *, b_rows=0, bias=None, act=ACT_NONE, residual=None, ldr=0,
          Cpre=None, colstats=None, ldstat=0, nbatch=1, inner=1, a_s=(0, 0), b_s=(0, 0), c_s=(0, 0),
          bias_s=(0, 0), omap=None, dropout=None, bnr=None, bna=None):
    """bnr = (y, z or None, mean, rstd, scale, shift, relu, partials): BatchNorm-backward sums of the consumer of Cout taken
    in the epilogue (pp_igemm_desc.bnr_*).  Returns True when the dispatched kernel took them (else run bn_bwd_reduce)."""
    d = IGemmDesc()
    if dropout is not None and dropout[0] > 0:     # (p, seed): the mask of dropout_bf16 on the flat output
        d.drop_p, d.drop_seed = float(dropout[0]), int(dropout[1]) & 0xffffffff
    d.M, d.N, d.K, d.g = M, N, K, g
    d.A, d.Bt, d.ldb, d.b_rows = _p(A, act16()), _p(Bt, act16()), ldb, b_rows
    d.C, d.ldc, d.c_fp32 = _p(Cout), ldc, int(Cout.dtype == f32)
    d.Cpre = _p(Cpre, act16())
    d.bias, d.act = _p(bias, f32), act
    d.residual, d.ldr = _p(residual, act16()), ldr
    d.colstats, d.ldstat = _p(colstats, f32), ldstat
    d.nbatch, d.inner = nbatch, inner
    d.a_s0, d.a_s1 = a_s
    d.b_s0, d.b_s1 = b_s
    d.c_s0, d.c_s1 = c_s
    d.bias_s0, d.bias_s1 = bias_s
    if omap is not None:  # ((Ot, Oh, Ow), (scale t,h,w), (offset t,h,w))
        d.omap = 1
        (d.Ot, d.Oh, d.Ow), (d.os_t, d.os_h, d.os_w), (d.oo_t, d.oo_h, d.oo_w) = omap
    if bna is not None:      # A is a BatchNorm unit's raw output y; the kernel reads relu?(y * scale + shift): (scale, shift, relu)
        d.a_bn_scale, d.a_bn_shift, d.a_bn_relu = _p(bna[0], f32), _p(bna[1], f32), int(bna[2])
    if bnr is not None:
        y, z, mean, rstd, scale, shift, relu, partials = bnr
        d.bnr_y, d.bnr_z = _p(y, act16()), _p(z, act16())
        d.bnr_mean, d.bnr_rstd, d.bnr_scale, d.bnr_shift = _p(mean, f32), _p(rstd, f32), _p(scale, f32), _p(shift, f32)
        d.bnr_relu, d.bnr_partials = int(relu), _p(partials, f32)
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(("igemm", _MODE_NAMES[g.mode], M, N, K, nbatch, (g.kt, g.kh, g.kw), (g.st, g.sh, g.sw), g.cg))
    rc = []
    # (a strided data gradient launched as one problem: a dx row only meets the taps of its parity class)
    _profiled(f"{_igemm_family(g)}<{_MODE_NAMES[g.mode]}> N={N} K={K}", 2.0 * M * N * K * nbatch / (g.st * g.sh * g.sw if g.mode == CONV_DGRAD else 1),
              lambda: rc.append(call("pp_igemm", C.byref(d), _s())))
    return bnr is not None and rc[0] == 0


def igemm_stat_rows(M, N, K, g, *, bna=False):
    """Rows of the BatchNorm partial-statistics buffer a plain conv-forward pp_igemm of these sizes writes (pp_igemm_stat_rows)."""
    d = IGemmDesc()
    d.M, d.N, d.K, d.g, d.b_rows, d.nbatch = M, N, K, g, N, 1
    d.colstats = 1          # (any non-null value: only asks "with statistics")
    if bna:
        d.a_bn_scale = d.a_bn_shift = 1
    return int(_lib.lib().pp_igemm_stat_rows(C.byref(d)))


def igemm_abn_supported(M, N, K, g, *, colstats=True):
    """Would pp_igemm take a plain conv-forward of these sizes with a producer BatchNorm applied to A (igemm(bna=...))?"""
    d = IGemmDesc()
    d.M, d.N, d.K, d.g, d.b_rows, d.nbatch = M, N, K, g, N, 1
    return bool(_lib.lib().pp_igemm_abn_supported(C.byref(d)))


def wgrad_xbn_supported(M, Ni, Kj, g, ldy):
    d = WGradDesc()
    d.M, d.Ni, d.Kj, d.g, d.ldy, d.nbatch = M, Ni, Kj, g, ldy, 1
    return bool(_lib.lib().pp_wgrad_xbn_supported(C.byref(d)))


def wgrad(X, dY, dW, M, Ni, Kj, g, ldy, ldw, *, msplit=0, nbatch=1, x_s=0, dy_s=0, dw_s=0, dbias=None, dbias_s=0, x_bn=None,
          ptr_table=None):
    """x_bn = (scale, shift, relu): X is a BatchNorm unit's raw output y, the kernel uses relu?(y * scale + shift).
    ptr_table: int64 device tensor [nbatch][4] = {X, dY, dW, dbias or 0} of a grouped launch (pp_wgrad_desc.ptr_table);
    X / dY / dW / dbias then only say dtype and "bias wanted"."""
    d = WGradDesc()
    if ptr_table is not None:
        if ptr_table.dtype != torch.int64 or not ptr_table.is_cuda or ptr_table.numel() != 4 * nbatch:
            raise PeppaHipError("wgrad: ptr_table must be an int64 device tensor of nbatch x 4 pointers")
        d.ptr_table = ptr_table.data_ptr()
    if x_bn is not None:
        d.x_bn_scale, d.x_bn_shift, d.x_bn_relu = _p(x_bn[0], f32), _p(x_bn[1], f32), int(x_bn[2])
    d.M, d.Ni, d.Kj, d.g = M, Ni, Kj, g
    d.X, d.dY, d.ldy, d.dW, d.ldw = _p(X, act16()), _p(dY, act16()), ldy, _p(dW, f32), ldw
    d.msplit, d.nbatch, d.x_s, d.dy_s, d.dw_s = msplit, nbatch, x_s, dy_s, dw_s
    d.dbias, d.dbias_s = _p(dbias, f32), dbias_s
    ws = None
    if DETERMINISTIC:       # slabs for the splits' partial tiles (pp_wgrad_desc.ws); nothing when the launch is not split
        need = _lib.lib().pp_wgrad_ws_floats(C.byref(d))
        if need > 0:
            ws = torch.empty(need, dtype=f32, device=dW.device)
            d.ws, d.ws_floats = ws.data_ptr(), need
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(("wgrad", _MODE_NAMES[g.mode], M, Ni, Kj, nbatch, (g.kt, g.kh, g.kw), (g.st, g.sh, g.sw), g.cg))
    _profiled(f"{_wgrad_family(g)}<{_MODE_NAMES[g.mode]}> Ni={Ni} Kj={Kj}", 2.0 * M * Ni * Kj * nbatch,
              lambda: call("pp_wgrad", C.byref(d), _s()))


# ---- weight preparation ----------------------------------------------------------------------
def prep_conv_weight(w, out, Co, Ci, taps, rows_out, cg, transpose_io=False, flip=False, scale=1.0):
    call("pp_prep_conv_weight", _p(w, f32), Co, Ci, taps, _p(out, act16()), rows_out, cg, int(transpose_io), int(flip),
         scale, _s())


def select_taps(w, out, rows, taps, cg, sel):
    arr = (C.c_int * len(sel))(*sel)
    call("pp_select_taps", _p(w, act16()), rows, taps, cg, arr, len(sel), _p(out, act16()), _s())


def unprep_conv_grad(g, dw, Co, Ci, taps, cg):
    call("pp_unprep_conv_grad", _p(g, f32), Co, Ci, taps, cg, _p(dw, f32), _s())


def cast_pad_2d(inp, out, rows, cols, ld_in, rows_out, ld_out, transpose=False, cols_out=None):
    call("pp_cast_pad_2d", _p(inp, f32), rows, cols, ld_in, _p(out, act16()), rows_out,
         ld_out if cols_out is None else cols_out, ld_out, int(transpose), _s())


def cast_f32_to_bf16(inp, out):
    call("pp_cast_f32_to_bf16", _p(inp, f32), _p(out, act16()), inp.numel(), _s())


def cast_bf16_to_f32(inp, out):
    call("pp_cast_bf16_to_f32", _p(inp, act16()), _p(out, f32), inp.numel(), _s())


def cast_pad_2d_multi(jobs, device):
    """jobs: [(in fp32, out, rows, cols, ld_in, rows_out, cols_out, ld_out, transpose, out_f32)] -> one launch.
    Returns the device table (keep it alive until the launch has run: stream-ordered, the caller holds it)."""
    rows = [[t_in.data_ptr(), t_out.data_ptr(), r, c, ldi, ro, co, ldo, int(tr), int(of)]
            for (t_in, t_out, r, c, ldi, ro, co, ldo, tr, of) in jobs]
    host = torch.tensor(rows, dtype=torch.int64).pin_memory()
    tab = host.to(device, non_blocking=True)
    biggest = max(j[5] * j[6] for j in jobs)
    call("pp_cast_pad_2d_multi", C.c_void_p(tab.data_ptr()), len(jobs), max(1, min(256, (biggest + 1023) // 1024)), _s())
    return tab, host


def prep_conv_weight_multi(jobs, device):
    """jobs: [(w fp32 [Co][Ci][taps], out, Co, Ci, taps, rows_out, cg, transpose_io)] -> one launch (pp_prep_conv_weight_multi).
    Returns the device table and its pinned source (the caller keeps them until the launch has run)."""
    import struct
    one = struct.unpack("<q", struct.pack("<fi", 1.0, 0))[0]       # {float scale = 1, int pad} as the table's int64 word
    rows, blk0 = [], 0
    for (w, out, Co, Ci, taps, rows_out, cg, tr) in jobs:
        assert w.dtype == f32 and w.is_contiguous() and out.dtype == act16() and cg % 8 == 0
        assert (rows_out >= Ci and cg >= Co) if tr else (rows_out >= Co and cg >= Ci), "pad too small"
        assert out.numel() == rows_out * taps * cg < 2 ** 31 and out.data_ptr() % 16 == 0
        rows.append([w.data_ptr(), out.data_ptr(), Co, Ci, taps, rows_out, cg, int(tr), 0, blk0, one])
        blk0 += (rows_out * taps * cg + 2047) // 2048
    host = torch.tensor(rows, dtype=torch.int64).pin_memory()
    tab = host.to(device, non_blocking=True)
    call("pp_prep_conv_weight_multi", C.c_void_p(tab.data_ptr()), len(jobs), blk0, _s())
    return tab, host


_COPY_TABLES = {}      # (tuple of (src, dst, n)) -> (device table, pinned source, blocks): gradient tensors mostly keep their addresses


def copy_f32_multi(dsts, srcs):
    """dst_i <- src_i for contiguous fp32 tensors of equal sizes, ONE launch (pp_copy_f32_multi) instead of one copy each."""
    key, blk0 = [], 0
    for d, s_ in zip(dsts, srcs):
        if d.dtype != f32 or s_.dtype != f32 or d.numel() != s_.numel() or not (d.is_contiguous() and s_.is_contiguous()):
            raise PeppaHipError("copy_f32_multi: contiguous fp32 tensors of equal sizes")
        key.append((s_.data_ptr(), d.data_ptr(), d.numel()))
    key = tuple(key)
    hit = _COPY_TABLES.get(key)
    if hit is None:
        rows = []
        for (sp, dp, n) in key:
            rows.append([sp, dp, n, blk0])
            blk0 += (n + 4095) // 4096
        host = torch.tensor(rows, dtype=torch.int64).pin_memory()
        tab = host.to(dsts[0].device, non_blocking=True)
        if len(_COPY_TABLES) > 256:
            _COPY_TABLES.clear()
        hit = _COPY_TABLES[key] = (tab, host, blk0)
    call("pp_copy_f32_multi", C.c_void_p(hit[0].data_ptr()), len(key), hit[2], _s())


def copy_2d_f32(inp, ld_in, out, ld_out, rows, cols):
    call("pp_copy_2d_f32", _p(inp, f32), ld_in, _p(out, f32), ld_out, rows, cols, _s())


def transpose_bf16(inp, in_bs, ld_in, out, out_bs, ld_out, nb, R, Ccols, inner=1, in_s1=0, out_s1=0, r_pad=0):
    call("pp_transpose_bf16", _p(inp, act16()), in_bs, ld_in, _p(out, act16()), out_bs, ld_out, nb, R, Ccols, inner,
         in_s1, out_s1, r_pad, _s())


def fill_f32(t, v):
    call("pp_fill_f32", _p(t, f32), float(v), t.numel(), _s())


def video_normalize_ndhwc(x, out, mean3, std3):
    """out [B*T*H*W][8], or [B*T*H*W][4] (paired-pixel stem, pp_prep_conv_weight_pairs) by the shape of `out`."""
    B, _, T, H, W = x.shape
    m = (C.c_float * 3)(*mean3)
    sd = (C.c_float * 3)(*std3)
    call("pp_video_normalize_ndhwc4" if out.shape[-1] == 4 else "pp_video_normalize_ndhwc", _p(x, f32), _p(out, act16()),
         B, T, H, W, m, sd, _s())


def stem_pairs_stat_rows(images, Hi):
    return int(_lib.lib().pp_stem_pairs_stat_rows(images, Hi))


def stem_pairs_fwd(x, wf, y, colstats, images, Hi, Wp, Co, ldc, ldstat):
    """The paired-pixel stem convolution as a window kernel (pp_stem_pairs_fwd)."""
    call("pp_stem_pairs_fwd", _p(x, act16()), _p(wf, act16()), _p(y, act16()), _p(colstats, f32), images, Hi, Wp, Co, ldc, ldstat, _s())


def stem_pairs_wgrad(x, dy, dw, images, Hi, Wp, Co, ldy):
    """dw fp32 [Co][28][8] += the paired-pixel stem's weight gradient (pp_stem_pairs_wgrad; dw zeroed by the caller)."""
    call("pp_stem_pairs_wgrad", _p(x, act16()), _p(dy, act16()), _p(dw, f32), images, Hi, Wp, Co, ldy, _s())


def prep_conv_weight_pairs(w, out, Co, Ci, kth, kw, pw):
    call("pp_prep_conv_weight_pairs", _p(w, f32), Co, Ci, kth, kw, pw, _p(out, act16()), _s())


def unprep_conv_grad_pairs(g, dw, Co, Ci, kth, kw, pw):
    call("pp_unprep_conv_grad_pairs", _p(g, f32), Co, Ci, kth, kw, pw, _p(dw, f32), _s())


def video_normalize_u8_ndhwc(x, out, mean3, std3):
    """x uint8 (B,T,H,W,3) contiguous -> out 16-bit [B*T*H*W][8]."""
    B, T, H, W, c = x.shape
    if c != 3 or x.dtype != torch.uint8 or not x.is_contiguous():
        raise PeppaHipError(f"uint8 video must be contiguous (B,T,H,W,3), got {tuple(x.shape)} {x.dtype}")
    m = (C.c_float * 3)(*mean3)
    sd = (C.c_float * 3)(*std3)
    call("pp_video_normalize_u8_ndhwc4" if out.shape[-1] == 4 else "pp_video_normalize_u8_ndhwc", x.data_ptr(),
         _p(out, act16()), B, T, H, W, m, sd, _s())


def collate_video_u8(table, n, Tmax, H, W, out):
    """table: int64 device tensor [n][2] = {frames pointer, T_i}; out fp32 (n,3,Tmax,H,W)."""
    call("pp_collate_video_u8", table.data_ptr(), n, Tmax, H, W, _p(out, f32), _s())


def collate_rows(table, n, row_bytes, out):
    """table: int64 device tensor [n][2] = {pointer, bytes}; out: any contiguous tensor of n * row_bytes bytes."""
    if out.numel() * out.element_size() != n * row_bytes or not out.is_contiguous():
        raise PeppaHipError("collate_rows: output size mismatch")
    call("pp_collate_rows", table.data_ptr(), n, row_bytes, out.data_ptr(), _s())


def maxpool3x3s2_fwd(x, y, N, Hh, W, Cp):
    call("pp_maxpool3x3s2_fwd", _p(x, act16()), _p(y, act16()), N, Hh, W, Cp, _s())


def maxpool3x3s2_bwd(x, dy, dx, N, Hh, W, Cp):
    call("pp_maxpool3x3s2_bwd", _p(x, act16()), _p(dy, act16()), _p(dx, act16()), N, Hh, W, Cp, _s())


# ---- batch norm -----------------------------------------------------------------------------------
def bn_finalize(partials, nblk, ldstat, count, Cn, Cp, gamma, beta, eps, momentum, rmean, rvar, mean, rstd,
                scale, shift, ws=None):
    call("pp_bn_finalize", _p(partials, f32), nblk, ldstat, count, Cn, Cp, _p(gamma, f32), _p(beta, f32), eps,
         momentum, _p(rmean, f32), _p(rvar, f32), _p(mean, f32), _p(rstd, f32), _p(scale, f32), _p(shift, f32),
         _p(ws, f32), _s())


def partials_sum(partials, nblk, ld, ws, out):
    call("pp_partials_sum", _p(partials, f32), nblk, ld, _p(ws, f32), _p(out, f32), _s())


def bn_eval_affine(gamma, beta, rmean, rvar, eps, Cn, Cp, scale, shift):
    call("pp_bn_eval_affine", _p(gamma, f32), _p(beta, f32), _p(rmean, f32), _p(rvar, f32), eps, Cn, Cp, _p(scale, f32),
         _p(shift, f32), _s())


def colstats_bf16(y, M, Cp, partials, nblk):
    call("pp_colstats_bf16", _p(y, act16()), M, Cp, _p(partials, f32), nblk, _s())


def bn_apply(y, scale, shift, res, relu, z, M, Cp):
    call("pp_bn_apply", _p(y, act16()), _p(scale, f32), _p(shift, f32), _p(res, act16()), int(relu), _p(z, act16()), M, Cp, _s())


def bn_bwd_reduce(dz, y, z, mean, rstd, scale, shift, relu, partials, nblk, M, Cp):
    call("pp_bn_bwd_reduce", _p(dz, act16()), _p(y, act16()), _p(z, act16()), _p(mean, f32), _p(rstd, f32), _p(scale, f32),
         _p(shift, f32), int(relu), _p(partials, f32), nblk, M, Cp, _s())


def bn_bwd_finalize(partials, nblk, count, Cn, Cp, gamma, rstd, dgamma, dbeta, coef, ws=None):
    """ws: fp32 [64 * 2 * Cp] or None (then one level, whatever nblk)"""
    assert ws is None or ws.numel() >= 64 * 2 * Cp
    call("pp_bn_bwd_finalize", _p(partials, f32), nblk, count, Cn, Cp, _p(gamma, f32), _p(rstd, f32),
         _p(dgamma, f32), _p(dbeta, f32), _p(coef, f32), _p(ws, f32), _s())


def bn_bwd_apply(dz, y, z, mean, rstd, coef, scale, shift, relu, dy, dres, M, Cp):
    call("pp_bn_bwd_apply", _p(dz, act16()), _p(y, act16()), _p(z, act16()), _p(mean, f32), _p(rstd, f32), _p(coef, f32),
         _p(scale, f32), _p(shift, f32), int(relu), _p(dy, act16()), _p(dres, act16()), M, Cp, _s())


# ---- elementwise ----------------------------------------------------------------------------------
def gelu_fwd(x, y):
    call("pp_gelu_fwd", _p(x, act16()), _p(y, act16()), x.numel(), _s())


def gelu_bwd(dy, x, dx):
    call("pp_gelu_bwd", _p(dy, act16()), _p(x, act16()), _p(dx, act16()), x.numel(), _s())


def gelu_bwd_dropout(dy, x, dx, p, seed):
    call("pp_gelu_bwd_dropout", _p(dy, act16()), _p(x, act16()), _p(dx, act16()), x.numel(), float(p), int(seed) & 0xffffffff, _s())


def add_bf16(a, b, out):
    call("pp_add_bf16", _p(a, act16()), _p(b, act16()), _p(out, act16()), a.numel(), _s())


def dropout_bf16(x, y, p, seed, res=None):
    call("pp_dropout_bf16", _p(x, act16()), _p(res, act16()), _p(y, act16()), x.numel(), float(p), int(seed) & 0xffffffff, _s())


def dropout_f32(x, y, p, seed):
    call("pp_dropout_f32", _p(x, f32), _p(y, f32), x.numel(), float(p), int(seed) & 0xffffffff, _s())


def colsum_bf16(x, M, N, ld, out):
    call("pp_colsum_bf16", _p(x, act16()), M, N, ld, _p(out, f32), _s())


# ---- layer norm / softmax -------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, y, mean, rstd, rows, D):
    call("pp_layernorm_fwd", _p(x, act16()), _p(gamma, f32), _p(beta, f32), eps, _p(y, act16()), _p(mean, f32),
         _p(rstd, f32), rows, D, _s())


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, ws=None):
    """ws: optional fp32 scratch [blocks][2][D] -> two-pass (deterministic) dgamma / dbeta instead of atomics."""
    call("pp_layernorm_bwd", _p(dy, act16()), _p(x, act16()), _p(gamma, f32), _p(mean, f32), _p(rstd, f32), _p(dx, act16()),
         _p(dgamma, f32), _p(dbeta, f32), rows, D, _p(ws, f32), 0 if ws is None else ws.shape[0], _s())


def attention_fwd(qkv, B, T, heads, scale, p, seed, ctx, lse):
    call("pp_attention_fwd", _p(qkv, act16()), B, T, heads, float(scale), float(p), int(seed) & 0xffffffff, _p(ctx, act16()),
         _p(lse, f32), _s())


def attention_bwd(qkv, ctx, lse, dctx, B, T, heads, scale, p, seed, dqkv):
    call("pp_attention_bwd", _p(qkv, act16()), _p(ctx, act16()), _p(lse, f32), _p(dctx, act16()), B, T, heads, float(scale),
         float(p), int(seed) & 0xffffffff, _p(dqkv, act16()), _s())


def softmax_fwd(S, lds, P, ldp, nb, T, scale):
    call("pp_softmax_fwd", _p(S, f32), lds, _p(P, act16()), ldp, nb, T, scale, _s())


def softmax_bwd(dP, lds, P, ldp, dS, nb, T, scale):
    call("pp_softmax_bwd", _p(dP, f32), lds, _p(P, act16()), ldp, _p(dS, act16()), nb, T, scale, _s())


# ---- wav2vec2 conv0 + groupnorm, weight norm --------------------------------------------------------
def conv0_stats(wave, B, L, T0, w, stats):
    call("pp_conv0_stats", _p(wave, f32), B, L, T0, _p(w, f32), _p(stats, f32), _s())


def conv0_apply(wave, B, L, T0, w, stats, gamma, beta, eps, out):
    call("pp_conv0_apply", _p(wave, f32), B, L, T0, _p(w, f32), _p(stats, f32), _p(gamma, f32), _p(beta, f32), eps,
         _p(out, act16()), _s())


def conv0_bwd_reduce(wave, B, L, T0, w, stats, gamma, beta, eps, dout, red):
    call("pp_conv0_bwd_reduce", _p(wave, f32), B, L, T0, _p(w, f32), _p(stats, f32), _p(gamma, f32), _p(beta, f32),
         eps, _p(dout, act16()), _p(red, f32), _s())


def conv0_bwd_apply(wave, B, L, T0, w, stats, gamma, beta, eps, dout, red, dw, dgamma, dbeta):
    ws = torch.empty(B * 512 * 10, dtype=f32, device=wave.device) if DETERMINISTIC else None   # per-clip partial rows of dw
    call("pp_conv0_bwd_apply", _p(wave, f32), B, L, T0, _p(w, f32), _p(stats, f32), _p(gamma, f32), _p(beta, f32),
         eps, _p(dout, act16()), _p(red, f32), _p(dw, f32), _p(dgamma, f32), _p(dbeta, f32), _p(ws, f32), _s())


def weightnorm_fwd(v, g, Co, Ci, Kk, norm, out):
    call("pp_weightnorm_fwd", _p(v, f32), _p(g, f32), Co, Ci, Kk, _p(norm, f32), _p(out, act16()), _s())


def weightnorm_bwd(dwt, v, g, norm, Co, Ci, Kk, dv, dg, dot_ws):
    call("pp_weightnorm_bwd", _p(dwt, f32), _p(v, f32), _p(g, f32), _p(norm, f32), Co, Ci, Kk, _p(dv, f32),
         _p(dg, f32), _p(dot_ws, f32), _s())


# ---- heads ------------------------------------------------------------------------------------------
def spatial_mean_fwd(x, out, B, T, HW, Cn, Cp):
    call("pp_spatial_mean_fwd", _p(x, act16()), _p(out, f32), B, T, HW, Cn, Cp, _s())


def spatial_mean_bwd(dout, dx, B, T, HW, Cn, Cp):
    call("pp_spatial_mean_bwd", _p(dout, f32), _p(dx, act16()), B, T, HW, Cn, Cp, _s())


def avgpool_tf_fwd(x, out, B, T, F, S):
    call("pp_avgpool_tf_fwd", _p(x, f32), B, T, F, S, _p(out, f32), _s())


def avgpool_tf_bwd(dout, dx, B, T, F, S):
    call("pp_avgpool_tf_bwd", _p(dout, f32), B, T, F, S, _p(dx, f32), _s())


def attnpool_fwd(x, B, T, Fdim, Hd, E, W1, b1, W2, b2, Wp, bp, hid, alpha, pooled, pre, out, normalize=True):
    call("pp_attnpool_fwd", _p(x, f32), B, T, Fdim, Hd, E, _p(W1, f32), _p(b1, f32), _p(W2, f32), _p(b2, f32),
         _p(Wp, f32), _p(bp, f32), int(normalize), _p(hid, f32), _p(alpha, f32), _p(pooled, f32), _p(pre, f32), _p(out, f32),
         _s())


def attnpool_ws_floats(B, T, Fdim, Hd, E):
    return call("pp_attnpool_ws_floats", B, T, Fdim, Hd, E)


def attnpool_bwd(dout, x, B, T, Fdim, Hd, E, W1, W2, Wp, hid, alpha, pooled, pre, out, dx, dW1, db1, dW2, db2, dWp,
                 dbp, ws, normalize=True):
    call("pp_attnpool_bwd", _p(dout, f32), _p(x, f32), B, T, Fdim, Hd, E, _p(W1, f32), _p(W2, f32), _p(Wp, f32),
         int(normalize), _p(hid, f32), _p(alpha, f32), _p(pooled, f32), _p(pre, f32), _p(out, f32), _p(dx, f32), _p(dW1, f32),
         _p(db1, f32), _p(dW2, f32), _p(db2, f32), _p(dWp, f32), _p(dbp, f32), _p(ws, f32), _s())


# ---- loss ------------------------------------------------------------------------------------------
def triplet_workspace_bytes(N, D):
    return call("pp_triplet_workspace_bytes", N, D)


def triplet_loss_fwd(V, A, margin, loss, ws):
    N, D = V.shape
    call("pp_triplet_loss_fwd", _p(V, f32), _p(A, f32), N, D, margin, _p(loss, f32), _p(ws), ws.numel() * ws.element_size(),
         _s())


def triplet_loss_hardest_fwd(V, A, margin, loss, ws):
    N, D = V.shape
    call("pp_triplet_loss_hardest_fwd", _p(V, f32), _p(A, f32), N, D, margin, _p(loss, f32), _p(ws), ws.numel() * ws.element_size(),
         _s())


def triplet_loss_bwd(V, A, dloss, ws, dV, dA):
    N, D = V.shape
    call("pp_triplet_loss_bwd", _p(V, f32), _p(A, f32), N, D, _p(dloss, f32), _p(ws), _p(dV, f32), _p(dA, f32), _s())


def recall_at_n(S, idx, correct, Nmax, out):
    """S fp32 (Nr, Nc); idx int32 (nsets, size) or None; correct uint8 (Nr, Nc) or None; out fp32 (nsets, Nmax, rows)."""
    Nr, Nc = S.shape
    nsets, size = (idx.shape if idx is not None else (1, Nr))
    call("pp_recall_at_n", _p(S, f32), Nr, Nc, S.stride(0), _p(idx, torch.int32), nsets, size, _p(correct, torch.uint8), Nmax,
         _p(out, f32), _s())


def cosine_matrix(U, V, out):
    ws = torch.empty((U.shape[0] + V.shape[0]) * U.shape[1], dtype=f32, device=U.device)
    call("pp_cosine_matrix", _p(U, f32), _p(V, f32), U.shape[0], V.shape[0], U.shape[1], _p(out, f32), _p(ws, f32), _s())


def contrastive_fwd(S, margin, loss):
    N = S.shape[0]
    ws = torch.empty(N * N + 3 * N, dtype=f32, device=S.device)
    call("pp_contrastive_fwd", _p(S, f32), N, margin, _p(loss, f32), _p(ws, f32), _s())


def cosine_matrix_bwd(U, V, dS, dU, dV):
    Nu, Nv, D = U.shape[0], V.shape[0], U.shape[1]
    ws = torch.empty(2 * (Nu + Nv) * D + Nu + Nv, dtype=f32, device=U.device)
    call("pp_cosine_matrix_bwd", _p(U, f32), _p(V, f32), Nu, Nv, D, _p(dS, f32), _p(dU, f32), _p(dV, f32), _p(ws, f32), _s())


def contrastive_bwd(S, margin, dloss, dS):
    N = S.shape[0]
    ws = torch.empty(3 * N + 1, dtype=f32, device=S.device)
    call("pp_contrastive_bwd", _p(S, f32), N, margin, _p(dloss, f32), _p(dS, f32), _p(ws, f32), _s())


def triplet_accuracy(a, p, n, discrete, out):
    call("pp_triplet_accuracy", _p(a, f32), _p(p, f32), _p(n, f32), a.shape[0], a.shape[1], int(discrete), _p(out, f32), _s())


# ---- optimizer -------------------------------------------------------------------------------------
def bertadam_step(tl, chunk_tensor, chunk_off, n_chunks, chunk, norms, lr, b1, b2, eps, wd, max_norm, lr_t=None, skip=None):
    # `norms` is scratch the C entry point cannot size-check: [n_tensors] squared norms, and in the deterministic mode
    # sumsq_kernel also writes one partial per chunk behind them (csrc/optim.hip) -- a short buffer would be a silent
    # out-of-bounds device write
    need = tl.n_tensors + (n_chunks if DETERMINISTIC else 0)
    if norms.numel() < need:
        raise PeppaHipError(f"pp_bertadam_step: `norms` holds {norms.numel()} floats, needs {need} "
                            f"(n_tensors{' + n_chunks in the deterministic mode' if DETERMINISTIC else ''})")
    call("pp_bertadam_step", C.byref(tl), _p(chunk_tensor, torch.int32), _p(chunk_off, torch.int64), n_chunks, chunk,
         _p(norms, f32), lr, b1, b2, eps, wd, max_norm, _p(lr_t, f32), _p(skip, f32), _s())


def grad_unscale_check(tl, chunk_tensor, chunk_off, n_chunks, chunk, scale, found_inf):
    call("pp_grad_unscale_check", C.byref(tl), _p(chunk_tensor, torch.int32), _p(chunk_off, torch.int64), n_chunks, chunk,
         _p(scale, f32), _p(found_inf, f32), _s())


def amp_update_scale(scale, tracker, found_inf, growth, backoff, interval):
    call("pp_amp_update_scale", _p(scale, f32), _p(tracker, torch.int32), _p(found_inf, f32), float(growth), float(backoff),
         int(interval), _s())


def make_tensor_list(ps, gs, ms, vs, device):
    """Device-resident pointer tables for pp_bertadam_step; returns (TensorList, keepalive)."""
    n = len(ps)
    for group in (ps, gs, ms, vs):          # a host pointer in the table is a GPU memory fault, not an exception
        for t in group:
            if not t.is_cuda or t.dtype != f32 or not t.is_contiguous():
                raise PeppaHipError("BertAdam tensors (parameters, gradients, next_m, next_v) must be contiguous "
                                    f"fp32 on the GPU; got {t.dtype} on {t.device}")
    # pinned + non_blocking: a pageable copy would block the host until the whole backward pass has drained
    host = torch.tensor([[t.data_ptr() for t in ps], [t.data_ptr() for t in gs], [t.data_ptr() for t in ms],
                         [t.data_ptr() for t in vs], [t.numel() for t in ps]], dtype=torch.int64).pin_memory()
    tab = host.to(device, non_blocking=True)
    tl = TensorList()
    tl.n_tensors = n
    base = tab.data_ptr()
    tl.p, tl.g, tl.m, tl.v, tl.numel = base, base + 8 * n, base + 16 * n, base + 24 * n, base + 32 * n
    return tl, tab
