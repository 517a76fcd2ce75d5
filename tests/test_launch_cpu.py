"""`python bench.py --gpus N` without a launcher (VERDICT r2 item 3 / ADVICE r2): the parent starts N child ranks itself
or refuses; it never reports N ranks after running one."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_spawn_ranks_runs_every_rank_over_gloo(tmp_path):
    from peppa_amd.launch import spawn_ranks
    out = tmp_path / "sum.txt"
    rc = spawn_ranks(os.path.join(ROOT, "tests", "_rank_echo.py"), [str(out)], 2, env=_env(), timeout=300)
    assert rc == 0
    assert out.read_text() == "3 2 127.0.0.1"        # 1 + 2 from two ranks, rendezvous on the loopback address


def test_spawn_ranks_refuses_inside_a_launch(monkeypatch):
    from peppa_amd.launch import spawn_ranks
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="already inside"):
        spawn_ranks("x.py", [], 2)


def test_launcher_command_is_the_drivers_line():
    from peppa_amd.launch import launcher_command
    cmd = launcher_command("bench.py", ["--gpus", "4"], 4, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-3].endswith("bench.py") and cmd[-2:] == ["--gpus", "4"]


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than two GPUs")
def test_bench_refuses_more_ranks_than_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "refusing" in p.stderr and "--gpus 2" in p.stderr
    assert not p.stdout.strip()                       # no record at all, in particular none that says n_gpus: 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_two_ranks_sharing_one_gpu():
    """The whole `--gpus 2` path on a one-GPU box: parent spawns torch.distributed.run, two ranks share cuda:0 over gloo
    (RCCL refuses two ranks per device), embedding all-gather + bucketed all-reduce run, rank 0 prints ONE line."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--frames", "8", "--size", "64", "--samples", "16000", "--no-cpu-baseline"],
                       env=_env(PEPPA_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-500:]      # stdout = the record, nothing else (no library banners)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["config"]["parallelism"].startswith("dp2")
    assert rec["value"] > 0
