"""HBM bytes of the WHOLE training step by kernel family: every launch of the step -- BatchNorm,
LayerNorm, elementwise, optimizer and torch's own fills included -- from the same two rocprofv3 PMC passes
tools/prof_traffic.py reads (FETCH_SIZE and WRITE_SIZE in separate `--pmc` runs with `--kernel-trace` only, over
tools/step_loop.py --isolated).  FETCH_SIZE is KiB and counts wide streaming reads at HALF on gfx950
(MI355X_MICROARCH.md, HBM / rocprofv3 section): doubled here; WRITE_SIZE (KiB) is taken as read.

    python tools/prof_step_traffic.py FETCH_DIR WRITE_DIR NSTEPS STEP_MS OUT.md OUT.json
NSTEPS = training steps inside each profiled run (warm-up included); STEP_MS = the un-profiled step time the total is held
against (bench.py's ms_per_step on the same box)."""
import collections
import csv
import glob
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from prof_families import FAMILIES

HBM_SUSTAINED = 6.3e12     # what the streaming kernels of this library reach, bytes/s
HBM_PEAK = 8.0e12


def family(name):
    for label, pat in FAMILIES:
        if re.search(pat, name):
            return label
    return "other"


def load(d, counter):
    trace = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    names, dur = {}, {}
    for r in csv.DictReader(open(trace)):
        i = int(r["Dispatch_Id"])
        names[i] = r["Kernel_Name"]
        dur[i] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    vals = collections.defaultdict(float)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                vals[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    return names, dur, vals


def main(fetch_dir, write_dir, nsteps, step_ms, out_md, out_json):
    nsteps, step_ms = int(nsteps), float(step_ms)
    nf, df, vf = load(fetch_dir, "FETCH_SIZE")
    nw, dw, vw = load(write_dir, "WRITE_SIZE")
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])      # launches, ns, read bytes, written bytes
    for i, name in nf.items():
        a = fam[family(name)]
        a[0] += 1
        a[1] += df[i]
        a[2] += vf.get(i, 0.0) * 1024 * 2
    for i, name in nw.items():
        fam[family(name)][3] += vw.get(i, 0.0) * 1024
    rows = sorted(((k, c / nsteps, ns / nsteps / 1e6, rd / nsteps, wr / nsteps) for k, (c, ns, rd, wr) in fam.items()),
                  key=lambda r: -(r[3] + r[4]))
    tot_rd, tot_wr, tot_ms = sum(r[3] for r in rows), sum(r[4] for r in rows), sum(r[2] for r in rows)
    total = tot_rd + tot_wr
    floor_ms = total / HBM_SUSTAINED * 1e3
    with open(out_md, "w") as f:
        f.write("# HBM bytes of one training step by kernel family (PMC, every launch)\n\n"
                f"`rocprofv3 --pmc FETCH_SIZE --kernel-trace` / `--pmc WRITE_SIZE --kernel-trace` over `tools/step_loop.py --isolated` "
                f"({nsteps} steps per run, all counted; one stream, one kernel at a time), grouped by kernel family "
                "(tools/prof_step_traffic.py).  read = FETCH_SIZE KiB x 2 (the gfx950 correction), written = WRITE_SIZE KiB; GB per step.\n\n"
                "| family | launches/step | ms/step (profiled, alone) | read GB | written GB | total GB | share | GB/s while running |\n"
                "|---|---|---|---|---|---|---|---|\n")
        for k, c, ms, rd, wr in rows:
            f.write(f"| {k} | {c:.0f} | {ms:.2f} | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {(rd + wr) / 1e9:.2f} | {(rd + wr) / total * 100:.1f} % | "
                    f"{(rd + wr) / ms / 1e6 if ms > 0 else 0:.0f} |\n")
        f.write(f"| **step** | | **{tot_ms:.2f}** | **{tot_rd / 1e9:.2f}** | **{tot_wr / 1e9:.2f}** | **{total / 1e9:.2f}** | 100 % | |\n\n")
        f.write(f"One step moves **{total / 1e9:.1f} GB** through HBM.  At the {HBM_SUSTAINED / 1e12:.1f} TB/s the streaming kernels sustain that is "
                f"**{floor_ms:.1f} ms** = **{floor_ms / step_ms * 100:.0f} %** of the {step_ms:.2f} ms step "
                f"({total / HBM_PEAK * 1e3:.1f} ms at the nominal {HBM_PEAK / 1e12:.0f} TB/s); the step as a whole averages "
                f"{total / (step_ms * 1e-3) / 1e9:.0f} GB/s = {total / (step_ms * 1e-3) / HBM_PEAK * 100:.0f} % of the nominal peak.\n")
    json.dump({"step_bytes": total, "read_bytes": tot_rd, "written_bytes": tot_wr, "hbm_floor_ms_at_6p3": floor_ms, "step_ms": step_ms,
               "frac_of_step_at_6p3": floor_ms / step_ms,
               "families": {k: {"launches": c, "ms_alone": ms, "read": rd, "written": wr} for k, c, ms, rd, wr in rows}},
              open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:7])
