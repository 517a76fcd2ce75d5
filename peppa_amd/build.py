"""Build peppa_amd/libpeppa_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpeppa_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    # dlopen in a child process: an unresolved symbol (e.g. a kernel stub the host pass dropped) must fail the build
    # here, not the first call on the GPU box
    subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({OUT!r})"], check=True)
    return OUT


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
