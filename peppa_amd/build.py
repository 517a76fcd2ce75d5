"""Build peppa_amd/libpeppa_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpeppa_hip.so")
# flavour -> (shared object, extra compile flags, object directory): the same sources, bf16 and IEEE-half operands
FLAVOURS = {"bf16": (OUT, [], "build"), "fp16": (os.path.join(HERE, "libpeppa_hip_f16.so"), ["-DPP_F16"], "build_f16")}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
# -fno-slp-vectorize: no packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which the SLP vectoriser
# forms from pairs of scalar float operations).  With them, pp_layernorm_bwd returned one row of dx computed from slightly
# wrong sums in about one launch of ten whenever MFMA-issuing waves of the register-staged GEMM / weight-gradient kernels
# shared its SIMDs; without them, never (tools/probe/ln_pk_repro.hip is the standalone reproducer).
# The MFMA kernels have no use for the packed forms and the streaming kernels are HBM-bound: the step time is unchanged.
# The flag is a CORRECTNESS requirement, so it is enforced rather than trusted: the flag list is part of every object's
# staleness stamp (a build/ directory that predates a flag change is rebuilt) and check_no_packed_fp32() disassembles the
# linked library (tests/test_host_cpu.py runs the same check on the shipped files).
# "-target-feature -packed-fp32-ops" takes the instructions away from the BACKEND: round 3's -fno-slp-vectorize alone left
# nine of them in the library (formed by the loop vectoriser in timepool_bwd_kernel, wn_apply_kernel, unscale_check_kernel;
# found by the check below the first time it ran), and explicit float2 arithmetic compiles to them under any -f flag.
# (-Xclang reaches the host pass as well, which prints "'-packed-fp32-ops' is not a recognized feature for this target":
# filtered below, every other compiler message is shown.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_HOST_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)"
PACKED_FP32 = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
# experiment macros of tools/probe/*.sh (ablations whose results are WRONG, kernel variants): a library built with any of
# them reports it through pp_experimental_build() and is refused by peppa_amd._lib unless PEPPA_ALLOW_EXPERIMENTAL=1
EXPERIMENT_MACROS = ("PP_WIN_ABLATE", "PP_TW_ABLATE", "PP_LN_VARIANT")


def hipcc_version():
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    return " | ".join(line.strip() for line in out.splitlines()[:2])


def _stamp(cmd):
    """What an object was built FROM besides its sources: the full command line and the compiler's version."""
    return hashlib.sha256((" ".join(cmd) + "\n" + hipcc_version()).encode()).hexdigest()


def _stale(target, deps, stamp=None):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    if stamp is not None:
        try:
            with open(target + ".stamp") as f:
                return f.read().strip() != stamp
        except OSError:
            return True
    return False


def code_objects(path, workdir):
    """Unbundle the gfx950 code objects of a linked library / object file into `workdir`; returns their paths."""
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=workdir)
    return sorted(glob.glob(local + ".*gfx950"))


def packed_fp32_instructions(path):
    """{mnemonic: count} of the packed-FP32 VALU instructions in every gfx950 code object of `path` (llvm-objdump -d)."""
    counts = {}
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(path, tmp)
        if not cos:
            raise RuntimeError(f"{path}: no gfx950 code object found")
        for co in cos:
            dis = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            for m in PACKED_FP32.finditer(dis):
                counts[m.group(0)] = counts.get(m.group(0), 0) + 1
    return counts


def check_no_packed_fp32(path):
    counts = packed_fp32_instructions(path)
    if counts:
        raise RuntimeError(f"{path} contains packed-FP32 instructions {counts}: it must be built with -fno-slp-vectorize and "
                           "without explicit float2 arithmetic (DESIGN.md section 7: wrong LayerNorm-backward rows beside MFMA waves)")


def build_library(force=False, verbose=True, flavour="bf16", extra_flags=(), out=None, objname=None, allow_experimental=False):
    """`extra_flags` / `out` / `objname`: variant builds of tools/probe/*.sh go to their OWN objects and shared object
    (PEPPA_HIP_LIB points the binding at one); the shipped library is never overwritten by an experiment."""
    dflt_out, extra, dflt_obj = FLAVOURS[flavour]
    out, objname = out or dflt_out, objname or dflt_obj
    extra = list(extra) + list(extra_flags)
    if any(m in f for f in extra_flags for m in EXPERIMENT_MACROS) and out == dflt_out:
        raise RuntimeError("experiment macros must not be built into the shipped library: pass out= / objname=")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(HERE, objname)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        stamp = _stamp(cmd)
        if force or _stale(obj, [src] + hdrs, stamp):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            sys.stderr.write("".join(l for l in r.stderr.splitlines(True) if _HOST_NOISE not in l))
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            with open(obj + ".stamp", "w") as f:
                f.write(stamp)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(os.path.join(objdir, "BUILD_INFO.json"), "w") as f:
            json.dump({"hipcc": hipcc_version(), "flags": FLAGS + extra, "library": os.path.basename(out)}, f, indent=1)
    # dlopen in a child process: an unresolved symbol (e.g. a kernel stub the host pass dropped) must fail the build
    # here, not the first call on the GPU box; a default build must not report experiment macros
    probe = (f"import ctypes; h = ctypes.CDLL({out!r}); e = h.pp_experimental_build(); "
             f"assert {bool(allow_experimental)!r} or e == 0, 'experiment macros in the library: %d' % e")
    subprocess.run([sys.executable, "-c", probe], check=True)
    check_no_packed_fp32(out)
    return out


def build_all(force=False, verbose=True):
    return [build_library(force, verbose, flavour) for flavour in FLAVOURS]


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
