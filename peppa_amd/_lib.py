"""ctypes binding of libpeppa_hip.so (C ABI declared in include/peppa_hip.h).

The library is the product's only compute path: there is no CPU or PyTorch fallback.
`lib()` raises `PeppaHipError` when the shared object is missing or a call fails.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpeppa_hip.so")
# one shared object per 16-bit operand type (same sources, include/peppa_hip.h `pp_dtype`)
# PEPPA_HIP_LIB / PEPPA_HIP_LIB_F16: another build of the same library (tools/probe/*.sh link their variants to /tmp and
# point the binding there; the shipped files are never overwritten by an experiment)
LIB_PATHS = {"bf16": os.environ.get("PEPPA_HIP_LIB", LIB_PATH),
             "fp16": os.environ.get("PEPPA_HIP_LIB_F16", os.path.join(_HERE, "libpeppa_hip_f16.so"))}
PRECISION = "bf16"    # which of the two `call` dispatches to (peppa_amd.hip.set_precision)


class PeppaHipError(RuntimeError):
    pass


class Gather(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mode", "lda", "Rt", "Rh", "Rw", "Gt", "Gh", "Gw", "kt", "kh", "kw",
        "st", "sh", "sw", "pt", "ph", "pw", "cg", "cstride")]


class IGemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("g", Gather),
        ("A", C.c_void_p), ("Bt", C.c_void_p), ("ldb", C.c_int), ("b_rows", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int), ("c_fp32", C.c_int), ("Cpre", C.c_void_p),
        ("bias", C.c_void_p), ("act", C.c_int), ("residual", C.c_void_p), ("ldr", C.c_int),
        ("colstats", C.c_void_p), ("ldstat", C.c_int), ("nbatch", C.c_int), ("inner", C.c_int),
        ("a_s0", C.c_longlong), ("a_s1", C.c_longlong), ("b_s0", C.c_longlong), ("b_s1", C.c_longlong),
        ("c_s0", C.c_longlong), ("c_s1", C.c_longlong), ("bias_s0", C.c_longlong), ("bias_s1", C.c_longlong),
        ("omap", C.c_int), ("Ot", C.c_int), ("Oh", C.c_int), ("Ow", C.c_int), ("os_t", C.c_int), ("os_h", C.c_int),
        ("os_w", C.c_int), ("oo_t", C.c_int), ("oo_h", C.c_int), ("oo_w", C.c_int),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint),
        ("bnr_y", C.c_void_p), ("bnr_z", C.c_void_p), ("bnr_mean", C.c_void_p), ("bnr_rstd", C.c_void_p),
        ("bnr_scale", C.c_void_p), ("bnr_shift", C.c_void_p), ("bnr_relu", C.c_int), ("bnr_partials", C.c_void_p),
        ("a_bn_scale", C.c_void_p), ("a_bn_shift", C.c_void_p), ("a_bn_relu", C.c_int)]


class WGradDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("Ni", C.c_int), ("Kj", C.c_int), ("g", Gather),
        ("X", C.c_void_p), ("dY", C.c_void_p), ("ldy", C.c_int), ("dW", C.c_void_p), ("ldw", C.c_int),
        ("msplit", C.c_int), ("nbatch", C.c_int),
        ("x_s", C.c_longlong), ("dy_s", C.c_longlong), ("dw_s", C.c_longlong),
        ("dbias", C.c_void_p), ("dbias_s", C.c_longlong),
        ("x_bn_scale", C.c_void_p), ("x_bn_shift", C.c_void_p), ("x_bn_relu", C.c_int),
        ("ptr_table", C.c_void_p), ("ws", C.c_void_p), ("ws_floats", C.c_longlong)]


class TensorList(C.Structure):
    _fields_ = [("n_tensors", C.c_int), ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p),
                ("v", C.c_void_p), ("numel", C.c_void_p)]


P, I, L, F, Z = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

# name -> argtypes (everything returns int status unless listed in _RESTYPE)
SIGNATURES = {
    "pp_version": [],
    "pp_dtype": [],
    "pp_experimental_build": [],
    "pp_grad_unscale_check": [C.POINTER(TensorList), P, P, I, I, P, P, P],
    "pp_amp_update_scale": [P, P, P, F, F, I, P],
    "pp_last_error": [],
    "pp_set_option": [C.c_char_p, I],
    "pp_igemm": [C.POINTER(IGemmDesc), P],
    "pp_wgrad": [C.POINTER(WGradDesc), P],
    "pp_wgrad_ws_floats": [C.POINTER(WGradDesc)],
    "pp_igemm_abn_supported": [C.POINTER(IGemmDesc)],
    "pp_igemm_stat_rows": [C.POINTER(IGemmDesc)],
    "pp_wgrad_xbn_supported": [C.POINTER(WGradDesc)],
    "pp_prep_conv_weight": [P, I, I, I, P, I, I, I, I, F, P],
    "pp_select_taps": [P, I, I, I, C.POINTER(I), I, P, P],
    "pp_unprep_conv_grad": [P, I, I, I, I, P, P],
    "pp_cast_pad_2d": [P, I, I, I, P, I, I, I, I, P],
    "pp_cast_pad_2d_multi": [P, I, I, P],
    "pp_prep_conv_weight_multi": [P, I, L, P],
    "pp_copy_f32_multi": [P, I, L, P],
    "pp_stem_pairs_fwd": [P, P, P, P, I, I, I, I, I, I, P],
    "pp_stem_pairs_stat_rows": [I, I],
    "pp_stem_pairs_wgrad": [P, P, P, I, I, I, I, I, P],
    "pp_cast_f32_to_bf16": [P, P, L, P],
    "pp_cast_bf16_to_f32": [P, P, L, P],
    "pp_copy_2d_f32": [P, I, P, I, I, I, P],
    "pp_transpose_bf16": [P, L, I, P, L, I, I, I, I, I, L, L, I, P],
    "pp_fill_f32": [P, F, L, P],
    "pp_video_normalize_ndhwc": [P, P, I, I, I, I, C.POINTER(F), C.POINTER(F), P],
    "pp_collate_video_u8": [P, I, I, I, I, P, P],
    "pp_collate_rows": [P, I, L, P, P],
    "pp_video_normalize_u8_ndhwc": [P, P, I, I, I, I, C.POINTER(F), C.POINTER(F), P],
    "pp_video_normalize_ndhwc4": [P, P, I, I, I, I, C.POINTER(F), C.POINTER(F), P],
    "pp_video_normalize_u8_ndhwc4": [P, P, I, I, I, I, C.POINTER(F), C.POINTER(F), P],
    "pp_prep_conv_weight_pairs": [P, I, I, I, I, I, P, P],
    "pp_unprep_conv_grad_pairs": [P, I, I, I, I, I, P, P],
    "pp_maxpool3x3s2_fwd": [P, P, I, I, I, I, P],
    "pp_maxpool3x3s2_bwd": [P, P, P, I, I, I, I, P],
    "pp_partials_sum": [P, I, I, P, P, P],
    "pp_bn_finalize": [P, I, I, L, I, I, P, P, F, F, P, P, P, P, P, P, P, P],
    "pp_colstats_bf16": [P, L, I, P, I, P],
    "pp_bn_eval_affine": [P, P, P, P, F, I, I, P, P, P],
    "pp_bn_apply": [P, P, P, P, I, P, L, I, P],
    "pp_bn_bwd_reduce": [P, P, P, P, P, P, P, I, P, I, L, I, P],
    "pp_bn_bwd_finalize": [P, I, L, I, I, P, P, P, P, P, P, P],
    "pp_bn_bwd_apply": [P, P, P, P, P, P, P, P, I, P, P, L, I, P],
    "pp_gelu_fwd": [P, P, L, P],
    "pp_gelu_bwd": [P, P, P, L, P],
    "pp_gelu_bwd_dropout": [P, P, P, L, F, C.c_uint, P],
    "pp_add_bf16": [P, P, P, L, P],
    "pp_dropout_bf16": [P, P, P, L, F, C.c_uint, P],
    "pp_dropout_f32": [P, P, L, F, C.c_uint, P],
    "pp_colsum_bf16": [P, L, I, I, P, P],
    "pp_layernorm_fwd": [P, P, P, F, P, P, P, I, I, P],
    "pp_layernorm_bwd": [P, P, P, P, P, P, P, P, I, I, P, I, P],
    "pp_attention_fwd": [P, I, I, I, F, F, C.c_uint, P, P, P],
    "pp_attention_bwd": [P, P, P, P, I, I, I, F, F, C.c_uint, P, P],
    "pp_softmax_fwd": [P, I, P, I, I, I, F, P],
    "pp_softmax_bwd": [P, I, P, I, P, I, I, F, P],
    "pp_conv0_stats": [P, I, I, I, P, P, P],
    "pp_conv0_apply": [P, I, I, I, P, P, P, P, F, P, P],
    "pp_conv0_bwd_reduce": [P, I, I, I, P, P, P, P, F, P, P, P],
    "pp_conv0_bwd_apply": [P, I, I, I, P, P, P, P, F, P, P, P, P, P, P, P],
    "pp_weightnorm_fwd": [P, P, I, I, I, P, P, P],
    "pp_weightnorm_bwd": [P, P, P, P, I, I, I, P, P, P, P],
    "pp_spatial_mean_fwd": [P, P, I, I, I, I, I, P],
    "pp_spatial_mean_bwd": [P, P, I, I, I, I, I, P],
    "pp_avgpool_tf_fwd": [P, I, I, I, I, P, P],
    "pp_avgpool_tf_bwd": [P, I, I, I, I, P, P],
    "pp_attnpool_fwd": [P, I, I, I, I, I, P, P, P, P, P, P, I, P, P, P, P, P, P],
    "pp_attnpool_bwd": [P, P, I, I, I, I, I, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "pp_attnpool_ws_floats": [I, I, I, I, I],
    "pp_triplet_workspace_bytes": [I, I],
    "pp_triplet_loss_fwd": [P, P, I, I, F, P, P, Z, P],
    "pp_triplet_loss_hardest_fwd": [P, P, I, I, F, P, P, Z, P],
    "pp_triplet_loss_bwd": [P, P, I, I, P, P, P, P, P],
    "pp_recall_at_n": [P, I, I, I, P, I, I, P, I, P, P],
    "pp_cosine_matrix": [P, P, I, I, I, P, P, P],
    "pp_contrastive_fwd": [P, I, F, P, P, P],
    "pp_cosine_matrix_bwd": [P, P, I, I, I, P, P, P, P, P],
    "pp_contrastive_bwd": [P, I, F, P, P, P, P],
    "pp_triplet_accuracy": [P, P, P, I, I, I, P, P],
    "pp_bertadam_step": [C.POINTER(TensorList), P, P, I, I, P, F, F, F, F, F, F, P, P, P],
}
_RESTYPE = {"pp_igemm_stat_rows": L, "pp_stem_pairs_stat_rows": L, "pp_last_error": C.c_char_p, "pp_attnpool_ws_floats": Z, "pp_triplet_workspace_bytes": Z, "pp_wgrad_ws_floats": L}
_NO_STATUS = set(_RESTYPE) | {"pp_version", "pp_dtype", "pp_experimental_build"}

_libs = {}
STICKY_OPTIONS = {}   # options set through peppa_amd.hip (e.g. deterministic): applied to a build that is loaded later


def lib(precision=None):
    """Load (once) and return the shared library of `precision` ("bf16" | "fp16", default: the current one); fail
    loudly if it is not built."""
    precision = PRECISION if precision is None else precision
    if precision not in _libs:
        path = LIB_PATHS[precision]
        if not os.path.exists(path):
            raise PeppaHipError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). peppa_amd has no CPU/PyTorch fallback.")
        h = C.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the header and the .so disagree
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        if h.pp_dtype() != {"bf16": 0, "fp16": 1}[precision]:
            raise PeppaHipError(f"{path} was not built for {precision} operands (pp_dtype = {h.pp_dtype()})")
        exp = h.pp_experimental_build()
        if exp and os.environ.get("PEPPA_ALLOW_EXPERIMENTAL") != "1":
            raise PeppaHipError(f"{path} was built with experiment macros (pp_experimental_build = {exp}: PP_WIN_ABLATE 1, "
                                "PP_TW_ABLATE 2, PP_LN_VARIANT 4): ablation / variant builds of tools/probe/, whose results "
                                "may be wrong by design.  Rebuild with peppa_amd.build, or set PEPPA_ALLOW_EXPERIMENTAL=1")
        _libs[precision] = h
        # tuning switches for A/B measurements: PEPPA_HIP_OPTIONS="ring_igemm=0,xcd_remap_wgrad=0"
        for item in filter(None, os.environ.get("PEPPA_HIP_OPTIONS", "").split(",")):
            key, _, val = item.partition("=")
            if h.pp_set_option(key.strip().encode(), int(val)) != 0:
                raise PeppaHipError(f"PEPPA_HIP_OPTIONS: {h.pp_last_error().decode()}")
        for key, val in STICKY_OPTIONS.items():
            if h.pp_set_option(key.encode(), int(val)) != 0:
                raise PeppaHipError(f"pp_set_option({key}): {h.pp_last_error().decode()}")
    return _libs[precision]


def call(name, *args):
    h = lib()
    rc = getattr(h, name)(*args)
    if name not in _NO_STATUS and rc < 0:
        raise PeppaHipError(f"{name} failed ({rc}): {h.pp_last_error().decode()}")
    return rc     # 0, or a positive "done, but ..." code (PP_BNR_SKIPPED)
