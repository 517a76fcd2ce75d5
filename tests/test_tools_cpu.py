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


def test_step_hbm_account_sums_every_family_with_the_gfx950_read_correction(tmp_path):
    """tools/prof_step_traffic.py: FETCH_SIZE KiB x 2 and WRITE_SIZE KiB per dispatch, grouped by
    kernel family, divided by the steps in the run; the total against the un-profiled step time."""
    fd, wd = tmp_path / "fetch" / "run", tmp_path / "write" / "run"
    fd.mkdir(parents=True)
    wd.mkdir(parents=True)
    rows, did = [], 1
    for step in range(2):
        for name, dur in ((WIN, 600.0), (BN, 400.0), ("bn_bwd_apply_kernel<true>(x)", 500.0), ("bertadam_kernel(pp_tensor_list)", 700.0)):
            rows.append((did, 1, name, did * 1000.0, did * 1000.0 + dur))
            did += 1
    for d in (fd, wd):
        _trace(str(d / "1_kernel_trace.csv"), rows)

    def counters(path, counter, kib):
        with open(path, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_ALL)
            w.writerow(["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id",
                        "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                        "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            for r in rows:      # two XCC partial rows per dispatch: the tool sums them
                for part in range(2):
                    w.writerow([r[0], r[0], 2, 1, 1, 1, 1, 1, r[2], 256, 0, 0, 64, 0, 32, counter, kib[r[2]] / 2, 0, 0])
    counters(str(fd / "1_counter_collection.csv"), "FETCH_SIZE", {WIN: 500000, BN: 200000, rows[2][2]: 400000, rows[3][2]: 100000})
    counters(str(wd / "1_counter_collection.csv"), "WRITE_SIZE", {WIN: 300000, BN: 200000, rows[2][2]: 200000, rows[3][2]: 50000})
    md, js = str(tmp_path / "acct.md"), str(tmp_path / "acct.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_step_traffic.py"), str(tmp_path / "fetch"), str(tmp_path / "write"),
                    "2", "40.0", md, js], check=True)
    acct = json.load(open(js))
    want_read = (500000 + 200000 + 400000 + 100000) * 1024 * 2
    want_written = (300000 + 200000 + 200000 + 50000) * 1024
    assert abs(acct["read_bytes"] - want_read) < 1 and abs(acct["written_bytes"] - want_written) < 1
    assert abs(acct["step_bytes"] - (want_read + want_written)) < 1
    assert abs(acct["families"]["BatchNorm apply"]["read"] - 200000 * 1024 * 2) < 1
    assert abs(acct["hbm_floor_ms_at_6p3"] - (want_read + want_written) / 6.3e12 * 1e3) < 1e-9
    assert "BatchNorm backward apply" in open(md).read()
