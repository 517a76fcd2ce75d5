"""The profile post-processing that `profiles/r03_shapes.md` / `r03_step_timeline` rest on: the join of a
rocprofv3 kernel trace with the launch log by dispatch order, its consistency checks, and the per-stream timeline."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HEADER = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
          "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
          "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]


def _trace(path, rows):
    """rows: (dispatch id, queue, name, start us, end us)"""
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(HEADER)
        for did, q, name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", "Agent 2", q, q - 1, 1, did, 6, name, did, int(s * 1000), int(e * 1000), 1024, 0, 128, 0, 32,
                        512, 1, 1, 131072, 1, 1])


WIN = "void (anonymous namespace)::igemm_win_kernel<9, 64, false, 2, 3, false, false, false, false>(WinArgs, WinGeom, int, int, int, int)"
WG = "void (anonymous namespace)::wgrad_kernel<8, 0, true>(pp_wgrad_desc, WGeom, int, int, int, int)"
BN = "bn_apply_kernel<false, true>(unsigned short const*)"


def _log(path, steps, warmup, calls):
    json.dump({"steps": steps, "warmup": warmup, "batch": 64, "frames": 16, "samples": 36800, "dtype": "bf16", "isolated": False,
               "launches": calls}, open(path, "w"))


def test_shapes_join_separates_shapes_that_share_a_kernel_template(tmp_path):
    import prof_shapes as PS
    big = ["igemm", "conv_fwd", 3211264, 144, 576, 1, [1, 3, 3], [1, 1, 1], 64]
    small = ["igemm", "conv_fwd", 401408, 288, 1152, 1, [1, 3, 3], [1, 1, 1], 128]
    wg = ["wgrad", "dense", 7296, 768, 768, 12, [1, 1, 1], [1, 1, 1], 768]
    rows, calls, t, did = [], [], 0.0, 1
    for step in range(3):                      # 1 warm-up + 2 timed steps; the same template runs two shapes
        for call, name, dur in ((big, WIN, 640.0), (None, BN, 50.0), (small, WIN, 270.0), (wg, WG, 200.0)):
            rows.append((did, 1, name, t, t + dur + step))
            if call is not None:
                calls.append(call)
            t += dur + 5
            did += 1
    trace, log = str(tmp_path / "t_kernel_trace.csv"), str(tmp_path / "launch.json")
    _trace(trace, rows)
    _log(log, 2, 1, calls)
    table, meta = PS.join(trace, log)
    keys = list(table)
    assert len(keys) == 3 and meta["steps"] == 2
    by_m = {k[2]: v for k, v in table.items()}
    assert by_m[3211264]["us"] == [641.0, 642.0] and by_m[401408]["us"] == [271.0, 272.0]      # warm-up step left out
    assert abs(PS.flops([k for k in keys if k[2] == 7296][0]) - 2.0 * 7296 * 768 * 768 * 12) < 1
    out = str(tmp_path / "shapes.md")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_shapes.py"), trace, log, out, "t"], check=True)
    text = open(out).read()
    assert "| 3211264 | 144 | 576 |" in text and "| 401408 | 288 | 1152 |" in text and "7296 x12" in text


def test_shapes_join_refuses_a_log_that_does_not_match_the_trace(tmp_path):
    import prof_shapes as PS
    trace, log = str(tmp_path / "t_kernel_trace.csv"), str(tmp_path / "launch.json")
    _trace(trace, [(1, 1, WIN, 0, 10), (2, 1, WG, 20, 30)])
    _log(log, 1, 0, [["igemm", "conv_fwd", 256, 64, 576, 1, [1, 3, 3], [1, 1, 1], 64]])
    with pytest.raises(SystemExit, match="2 matrix-kernel dispatches but 1 logged"):
        PS.join(trace, log)
    _log(log, 1, 0, [["wgrad", "dense", 256, 64, 64, 1, [1, 1, 1], [1, 1, 1], 64], ["igemm", "dense", 256, 64, 64, 1, [1, 1, 1], [1, 1, 1], 64]])
    with pytest.raises(SystemExit, match="the join is off"):
        PS.join(trace, log)


def test_stream_timeline_splits_idle_time_by_cause(tmp_path):
    trace, out = str(tmp_path / "t_kernel_trace.csv"), str(tmp_path / "tl.md")
    rows, did = [], 1
    for step in range(3):
        t0 = step * 1000.0
        rows += [(did, 1, WIN, t0 + 0, t0 + 400), (did + 1, 2, WG, t0 + 100, t0 + 700), (did + 2, 1, BN, t0 + 650, t0 + 800),
                 (did + 3, 1, "bertadam_kernel(pp_tensor_list)", t0 + 900, t0 + 950)]
        did += 4
    _trace(trace, rows)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_timeline.py"), trace, out, "t", "--steps", "2"], check=True)
    text = open(out).read()
    assert "1000.00" not in text and "**1.00 ms per step**" in text
    line = [l for l in text.splitlines() if l.startswith("| 1/0")][0].split("|")
    # trunk stream: busy 0.4 + 0.15 + 0.05 ms; gap 400..650 while stream 2 runs, gap 800..900 with nothing running
    assert abs(float(line[3]) - 0.60) < 1e-6 and abs(float(line[6]) - 0.25) < 1e-6 and abs(float(line[7]) - 0.15) < 1e-6
