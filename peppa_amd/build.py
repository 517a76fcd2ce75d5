"""Build peppa_amd/libpeppa_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpeppa_hip.so")
# flavour -> (shared object, extra compile flags, object directory): the same sources, bf16 and IEEE-half operands
FLAVOURS = {"bf16": (OUT, [], "build"), "fp16": (os.path.join(HERE, "libpeppa_hip_f16.so"), ["-DPP_F16"], "build_f16")}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: no packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which the SLP vectoriser
# forms from pairs of scalar float operations).  With them, pp_layernorm_bwd returned one row of dx computed from slightly
# wrong sums in about one launch of ten whenever waves of the register-staged GEMM / weight-gradient kernels shared its CUs;
# without them, never (tools/probe/ln_variants.sh with EXTRA=-fno-slp-vectorize).  The MFMA kernels
# have no use for the packed forms and the streaming kernels are HBM-bound: the step time is unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, flavour="bf16"):
    out, extra, objname = FLAVOURS[flavour]
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(HERE, objname)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # dlopen in a child process: an unresolved symbol (e.g. a kernel stub the host pass dropped) must fail the build
    # here, not the first call on the GPU box
    subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({out!r})"], check=True)
    return out


def build_all(force=False, verbose=True):
    return [build_library(force, verbose, flavour) for flavour in FLAVOURS]


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
