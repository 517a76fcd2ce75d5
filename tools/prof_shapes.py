"""Per-SHAPE kernel table from a rocprofv3 kernel trace: rocprofv3's own statistics aggregate by kernel
template, and the persistent kernels launch the same grid for every shape, so `igemm_win_kernel<9,64,...>` mixes the
layer-1/2/3/4 convolutions.  Every pp_igemm / pp_wgrad call is exactly ONE kernel launch; tools/step_loop.py --launch-log
records the calls (mode, M, N, K, taps, strides) in host order, which is the trace's Dispatch_Id order.  This script joins the
two (checking the count and, row by row, that an igemm call met an igemm kernel and a wgrad call a wgrad kernel) and prints
one row per (kernel template, shape): launches per step, avg / min / max microseconds, TFLOP/s, fraction of the dense peak.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/step_loop.py --steps 6 --launch-log DIR/launch.json
    python tools/prof_shapes.py DIR/*/*_kernel_trace.csv DIR/launch.json OUT.md "title" [--pmc PMC_DIR ...] [--alone TRACE LOG]

--alone: a second (trace, log) pair from `step_loop.py --isolated` adds the kernel's duration alone on the GPU.
--pmc: counter_collection.csv files of `rocprofv3 --pmc ...` runs of the same command (one pass per counter group) add
       MFMA-pipe utilisation and the wave-cycle split (see prof_pmc_columns).
"""
import argparse
import collections
import csv
import glob
import json
import re
import sys

PEAK = 2.5e15
GEMM_RE = re.compile(r"\b(igemm_win_kernel|igemm_kernel|wgrad_kernel|wgrad_big_kernel|wgrad_sw_kernel|wgrad_tw_kernel|wgrad_ring_kernel|wgrad_group_kernel)<")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+<[^>]*>)", name)
    return m.group(1).replace(" ", "") if m else name.split("(")[0]


def load_trace(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if GEMM_RE.search(r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]),
                         int(r["LDS_Block_Size"])))
    rows.sort()
    return rows


def join(trace_path, log_path):
    rows = load_trace(trace_path)
    log = json.load(open(log_path))
    calls = log["launches"]
    if len(rows) != len(calls):
        raise SystemExit(f"{trace_path}: {len(rows)} matrix-kernel dispatches but {len(calls)} logged pp_igemm/pp_wgrad calls")
    per_step = len(calls) // (log["steps"] + log["warmup"])
    out = collections.OrderedDict()
    first_timed = log["warmup"] * per_step          # warm-up steps are left out of the statistics
    for i, (row, call) in enumerate(zip(rows, calls)):
        entry = call[0]
        if not row[1].startswith(entry):
            raise SystemExit(f"dispatch {row[0]} is {row[1]} but call {i} was pp_{entry}: the join is off")
        if i < first_timed:
            continue
        key = (row[1], call[1], call[2], call[3], call[4], call[5], tuple(call[6]), tuple(call[7]))
        out.setdefault(key, {"us": [], "wgs": row[3], "regs": row[4], "lds": row[5], "ids": []})
        out[key]["us"].append(row[2])
        out[key]["ids"].append(row[0])
    return out, log


def flops(key):
    """Algorithmic FLOPs of one launch.  A strided data gradient launched as ONE problem (the window kernel's S2D form, the
    masked gather form) has M = rows of dx and K = every tap, but a dx row only meets the taps of its parity class:
    st * sh * sw times fewer products than M * N * K."""
    _, mode, M, N, K, nb, _, stride = key
    f = 2.0 * M * N * K * nb
    if mode == "conv_dgrad":
        f /= stride[0] * stride[1] * stride[2]
    return f


def load_pmc(dirs):
    """{dispatch id: {counter: value}} summed over the csv files of all passes (each pass re-runs the same command, so
    Dispatch_Id numbers line up across passes)."""
    vals = collections.defaultdict(dict)
    for d in dirs:
        for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                vals[int(r["Dispatch_Id"])][r["Counter_Name"]] = vals[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("log")
    ap.add_argument("out")
    ap.add_argument("title")
    ap.add_argument("--alone", nargs=2, default=None)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--pmc-log", default=None, help="launch log of the --pmc runs (default: the main log)")
    ap.add_argument("--min-ms", type=float, default=0.02)
    args = ap.parse_args()
    table, log = join(args.trace, args.log)
    alone = join(*args.alone)[0] if args.alone else {}
    # rocprofv3 stats of a PMC run are per dispatch: map dispatch -> shape through the same join on the PMC run's own trace
    pmc_by_key = {}
    if args.pmc:
        vals = load_pmc(args.pmc)
        ptrace = glob.glob(f"{args.pmc[0]}/**/*kernel_trace.csv", recursive=True)
        ptab, _ = join(ptrace[0], args.pmc_log or args.log)
        for key, rec in ptab.items():
            agg = collections.defaultdict(float)
            for did in rec["ids"]:
                for c, v in vals.get(did, {}).items():
                    agg[c] += v
            pmc_by_key[key] = {c: v / len(rec["ids"]) for c, v in agg.items()}
    steps = log["steps"]
    rows = sorted(table.items(), key=lambda kv: -sum(kv[1]["us"]))
    tot = sum(sum(v["us"]) for _, v in rows) / steps / 1e3
    with open(args.out, "w") as f:
        f.write(f"# {args.title}\n\n")
        f.write(f"source: `{args.trace}` joined with `{args.log}` (tools/prof_shapes.py); {steps} timed steps after "
                f"{log['warmup']} warm-up, batch {log['batch']}, {log['frames']} frames, {log['samples']} samples, {log['dtype']}"
                f"{', one stream (kernels alone)' if log.get('isolated') else ', streams overlapped as bench.py runs them'}.  "
                f"Matrix-core launches: **{tot:.2f} ms/step** of kernel time.  TF/s = 2*M*N*K / avg; frac = TF/s / 2500.\n\n")
        hdr = "| kernel | mode | M | N | K | taps | stride | n/step | avg us | min | max | TF/s | frac |"
        sep = "|---|---|---|---|---|---|---|---|---|---|---|---|---|"
        if alone:
            hdr, sep = hdr + " alone avg us | alone frac |", sep + "---|---|"
        if pmc_by_key:
            hdr, sep = hdr + " MFMA busy | waves: parked / stalled / issuing |", sep + "---|---|"
        f.write(hdr + "\n" + sep + "\n")
        for key, rec in rows:
            us = rec["us"]
            ms_step = sum(us) / steps / 1e3
            if ms_step < args.min_ms:
                continue
            avg = sum(us) / len(us)
            tf = flops(key) / (avg * 1e-6) / 1e12
            nb = f" x{key[5]}" if key[5] > 1 else ""
            line = (f"| `{key[0]}` | {key[1]} | {key[2]}{nb} | {key[3]} | {key[4]} | {'x'.join(map(str, key[6]))} | "
                    f"{'x'.join(map(str, key[7]))} | {len(us) / steps:.1f} | {avg:.1f} | {min(us):.1f} | {max(us):.1f} | {tf:.0f} | {tf * 1e12 / PEAK:.3f} |")
            if alone:
                a = alone.get(key)
                if a:
                    aavg = sum(a["us"]) / len(a["us"])
                    line += f" {aavg:.1f} | {flops(key) / (aavg * 1e-6) / PEAK:.3f} |"
                else:
                    line += " – | – |"
            if pmc_by_key:
                c = pmc_by_key.get(key, {})
                if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that were busy; SQ_BUSY_CYCLES is per
                    # shader engine (32 of them), GRBM_GUI_ACTIVE per XCD: utilisation = busy / (4 SIMDs x 256 CUs x cycles)
                    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
                    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else float("nan")
                    w = c.get("SQ_WAVE_CYCLES", 0.0) or float("nan")
                    line += (f" {util:.2f} | {c.get('SQ_WAIT_ANY', 0) / w:.2f} / {c.get('SQ_WAIT_INST_ANY', 0) / w:.2f} / "
                             f"{c.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} |")
                else:
                    line += " – | – |"
            f.write(line + "\n")
    print(f"wrote {args.out}: {len(rows)} shapes, {tot:.2f} ms/step")


if __name__ == "__main__":
    main()
