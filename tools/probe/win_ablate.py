"""Where the window kernels' time goes: rebuild igemm_win.hip with -DPP_WIN_ABLATE=<bits> (results wrong, timing only),
link it with the shipped objects into a scratch library and time the layer-1 convolutions with it.
Run on the GPU box:  python tools/probe/win_ablate.py [bits ...]   (default: a standard set)"""
import os, subprocess, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
NAMES = {1: "weights once", 2: "windows once", 4: "no reads/MFMA", 8: "no epilogue", 16: "reads, no MFMA", 32: "no global stores"}


def describe(bits):
    return " + ".join(v for k, v in NAMES.items() if bits & k) or "shipped"


def build(bits, tag=""):
    from peppa_amd import build as B
    B.build_library(verbose=False)
    objdir = os.path.join(ROOT, "peppa_amd", "build")
    out_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pp_abl")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"igemm_win_{bits}{tag}.o")
    lib = os.path.join(out_dir, f"libpeppa_abl_{bits}{tag}.so")
    extra = os.environ.get("ABL_FLAGS", "").split()
    subprocess.run([B.HIPCC] + B.FLAGS + extra + [f"-DPP_WIN_ABLATE={bits}", "-c", os.path.join(B.CSRC, "igemm_win.hip"), "-o", obj], check=True)
    others = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "igemm_win.o"]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
    return lib


def measure(lib_path):
    import torch
    from peppa_amd import _lib
    _lib.LIB_PATHS["bf16"] = lib_path
    from peppa_amd import layers as L

    def timeit(fn, n=10):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    out = []
    for name, Ci, Co, k, p in (("spatial 64->144", 64, 144, (1, 3, 3), (0, 1, 1)), ("temporal 144->64", 144, 64, (3, 1, 1), (1, 0, 0))):
        geom = L.ConvGeom(64, (16, 56, 56), Ci, Co, k, (1, 1, 1), p)
        x = torch.randn(geom.Min, geom.in_cstride, device="cuda").to(torch.bfloat16)
        dy = torch.randn(geom.M, geom.out_cstride, device="cuda").to(torch.bfloat16)
        wf, wd = L.prep_conv_weights(torch.randn(Co, Ci, *k, device="cuda") * 0.05, geom)
        out.append(f"{name}: fwd {timeit(lambda: L.conv_fwd(x, geom, wf, stats=True)):7.1f} us, dgrad {timeit(lambda: L.conv_dgrad(dy, geom, wd)):7.1f} us")
    print(" | ".join(out), flush=True)


def stamps(lib_path):
    """Per-segment cycle sums of a bits & 64 build: mean over workgroups, weight wave 0 and window wave 4."""
    import ctypes, numpy as np, torch
    from peppa_amd import _lib
    _lib.LIB_PATHS["bf16"] = lib_path
    from peppa_amd import layers as L
    h = _lib.lib()
    segs = ("wait", "barrier", "issue", "multiply", "turn", "epilogue", "total")
    for name, Ci, Co, k, p in (("spatial 64->144", 64, 144, (1, 3, 3), (0, 1, 1)), ("temporal 144->64", 144, 64, (3, 1, 1), (1, 0, 0))):
        geom = L.ConvGeom(64, (16, 56, 56), Ci, Co, k, (1, 1, 1), p)
        x = torch.randn(geom.Min, geom.in_cstride, device="cuda").to(torch.bfloat16)
        dy = torch.randn(geom.M, geom.out_cstride, device="cuda").to(torch.bfloat16)
        wf, wd = L.prep_conv_weights(torch.randn(Co, Ci, *k, device="cuda") * 0.05, geom)
        for what, fn in (("fwd", lambda: L.conv_fwd(x, geom, wf, stats=True)), ("dgrad", lambda: L.conv_dgrad(dy, geom, wd))):
            fn(); fn(); torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record(); fn(); t1.record(); torch.cuda.synchronize()
            buf = np.zeros((256, 8, 8), dtype=np.uint64)
            assert h.pp_debug_win_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
            us = t0.elapsed_time(t1) * 1e3
            line = [f"{name} {what}: {us:.0f} us; {buf[:, :, 6].mean() / us:.0f} ticks/us"]
            for w in (0, 4):
                m = buf[:, w, :].astype(np.float64).mean(axis=0)
                line.append(f"wave {w}: " + " ".join(f"{n} {100 * m[i] / m[6]:.1f}%" for i, n in enumerate(segs[:6])))
            print("\n    ".join(line), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--measure":
        measure(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--stamps-of":
        stamps(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--flags":       # A/B of -D switches: --flags "" "-DPP_WIN_SLACK=2" ...
        for rep in range(2):
            for i, fl in enumerate(sys.argv[2:]):
                os.environ["ABL_FLAGS"] = fl
                lib = build(0, f"_f{i}")
                print(f"[{fl or 'default':30s}]", end=" ", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--measure", lib], check=True)
    elif len(sys.argv) > 1 and sys.argv[1] == "--stamps":
        for bits in [int(a) | 64 for a in sys.argv[2:]] or [64]:
            lib = build(bits)
            print(f"[{bits:2d}] stamps, {describe(bits & 63)}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--stamps-of", lib], check=True)
    else:
        todo = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 8, 32, 4, 16, 12, 9, 11, 15]
        for bits in todo:
            lib = build(bits)
            print(f"[{bits:2d}] {describe(bits):45s}", end=" ", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--measure", lib], check=True)
