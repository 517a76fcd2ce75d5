"""Start one process per GPU for a script that was called without a launcher (`python bench.py --gpus N`).

The reference is single-GPU; the data-parallel step is new work and its ranks normally come from
`python -m torch.distributed.run`.  When WORLD_SIZE is unset and more than one GPU is asked for, the parent process --
which must not have touched the GPU yet (no HIP call, no `torch.cuda.is_available()`; `torch.cuda.device_count()` is
safe on this image) -- starts `torch.distributed.run` as a CHILD process, relays its output and exits with its code.
It never re-execs itself: replacing a process that has initialised the GPU takes the machine down on this pool.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(script, script_args, nproc, port=None):
    """argv of the child launcher: the same line the driver uses for N > 1 (rendezvous on 127.0.0.1)."""
    port = free_port() if port is None else port
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(script_args)


def spawn_ranks(script, script_args, nproc, env=None, timeout=None):
    """Run `script` as `nproc` ranks in child processes; returns the launcher's exit code.  stdout / stderr are
    inherited, so rank 0's one JSON line appears on the caller's stdout exactly as if the script had printed it."""
    if "WORLD_SIZE" in os.environ:
        raise RuntimeError("spawn_ranks: already inside a torch.distributed launch (WORLD_SIZE is set)")
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    child_env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(script, script_args, nproc)
    print(f"[launch] {nproc} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=child_env)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()      # the exact process we started (never a pattern kill)
        proc.wait()
        return 124


def require_devices(nproc, what="bench.py"):
    """Refuse -- loudly, non-zero -- to run fewer ranks than asked for.  PEPPA_BENCH_SHARE_GPU=1 is the one-GPU rehearsal
    of the multi-rank path: every rank uses cuda:0 and the collectives go through gloo (RCCL refuses two ranks on one
    device); its record says so and is never a scaling number."""
    import torch
    have = torch.cuda.device_count()     # (does not initialise the GPU on this image)
    if os.environ.get("PEPPA_BENCH_SHARE_GPU") == "1":
        if have < 1:
            raise SystemExit(f"{what}: PEPPA_BENCH_SHARE_GPU=1 needs at least one GPU, none is visible")
        return have
    if have < nproc:
        raise SystemExit(f"{what}: --gpus {nproc} asked for but {have} GPU(s) are visible; refusing to run fewer ranks "
                         f"and report them as {nproc} (start it on a node with {nproc} GPUs)")
    return have
