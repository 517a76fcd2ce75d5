"""Kernel families alone vs in-step: merges the kernel_stats.csv of two rocprofv3 runs of
tools/step_loop.py -- default (three streams overlapped) and --isolated (one kernel at a time) -- into one markdown table.

    python tools/prof_families.py IN_STEP.csv ISOLATED.csv NSTEPS OUT.md "title" [SKIP_STEPS]
"""
import csv
import re
import sys

FAMILIES = [
    ("window conv fwd/dgrad (igemm_win)", r"igemm_win_kernel"),
    ("stem window kernels (fwd + wgrad)", r"stem_pairs"),
    ("generic / ring implicit GEMM (igemm)", r"igemm_kernel"),
    ("sliding-window wgrad (wgrad_sw)", r"wgrad_sw_kernel"),
    ("temporal-window wgrad (wgrad_tw)", r"wgrad_tw_kernel"),
    ("generic wgrad (incl. grouped, 256 x 256 tiles)", r"wgrad_kernel|wgrad_big_kernel|wgrad_slab"),
    ("BatchNorm apply", r"bn_apply_kernel"),
    ("BatchNorm backward reduce", r"bn_bwd_reduce_kernel"),
    ("BatchNorm backward apply", r"bn_bwd_apply_kernel"),
    ("BatchNorm finalize / partials", r"bn_finalize|bn_bwd_finalize|partials_reduce|bn_eval"),
    ("attention fwd", r"attention_fwd"),
    ("attention bwd", r"attention_bwd"),
    ("LayerNorm", r"ln_"),
    ("wav2vec2 conv0 (+GroupNorm, GELU)", r"conv0_"),
    ("GELU / dropout / add / transpose", r"gelu|dropout|add_|transpose|colsum|softmax"),
    ("weight preparation (casts, layouts)", r"prep_conv|unprep_conv|cast_|select_taps|weightnorm|wn_"),
    ("heads + loss (fp32)", r"sgemm|timepool|l2norm|cosnorm|loss_|hinge|spatial_mean|copy_f32|diag_|gdiag"),
    ("BertAdam", r"bertadam|sumsq|unscale"),
    ("input normalise / collate / maxpool", r"video_normalize|collate|maxpool"),
    ("torch fills / copies", r"at::native|Memset|Memcpy|fill"),
]


def step_of(rows):
    """Training-step index of every kernel_trace row (in start order): a step ends with its bertadam_kernel launch; the first
    launch after it opens the next step."""
    order = sorted(range(len(rows)), key=lambda i: int(rows[i]["Start_Timestamp"]))
    idx, step, in_opt = [0] * len(rows), 0, False
    for i in order:
        opt = "bertadam_kernel" in rows[i]["Kernel_Name"]
        if in_opt and not opt:
            step += 1
        in_opt = opt
        idx[i] = step
    return idx


def rows_of(path, skip=0):
    """kernel_stats.csv rows (Name, Calls, TotalDurationNs), or the same aggregated from a kernel_trace.csv
    (skip: leave out the first `skip` training steps -- the very first one runs without the pooled clears and the
    cached weight operands)"""
    rows = list(csv.DictReader(open(path)))
    if rows and "Kernel_Name" in rows[0]:
        if skip:
            st = step_of(rows)
            rows = [r for r, k in zip(rows, st) if k >= skip]
        agg = {}
        for r in rows:
            a = agg.setdefault(r["Kernel_Name"], [0, 0])
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        return [{"Name": k, "Calls": c, "TotalDurationNs": ns} for k, (c, ns) in agg.items()]
    return rows


def load(path, steps, skip=0):
    fam = {}
    for r in rows_of(path, skip):
        name = r["Name"]
        for label, pat in FAMILIES:
            if re.search(pat, name):
                break
        else:
            label = "other"
        a = fam.setdefault(label, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    return {k: (c / steps, ns / steps / 1e6) for k, (c, ns) in fam.items()}


def main(in_step, isolated, steps, out, title, skip=0):
    """steps = training steps in each trace; skip = how many of them to leave out from the start (kernel traces only)"""
    a, b = load(in_step, steps - skip, skip), load(isolated, steps - skip, skip)
    ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsources: `{in_step}` (towers and weight gradients on three streams, as bench.py runs) and "
                f"`{isolated}` (same step, one stream, one kernel at a time); {steps} steps each"
                + (f", the first {skip} left out (step 0 runs without the pooled clears and cached operands)" if skip else "") + ", ms per step.\n\n")
        f.write("| family | launches/step | ms/step alone | ms/step in-step | in-step / alone |\n|---|---|---|---|---|\n")
        for label in sorted(set(a) | set(b), key=lambda k: -b.get(k, (0, 0))[1]):
            ca, ma = a.get(label, (0, 0.0))
            cb, mb = b.get(label, (0, 0.0))
            f.write(f"| {label} | {max(ca, cb):.1f} | {mb:.2f} | {ma:.2f} | {ma / mb if mb else float('nan'):.2f} |\n")
        f.write(f"| **total kernel time** | | **{tb:.2f}** | **{ta:.2f}** | {ta / tb:.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 0)
