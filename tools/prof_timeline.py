"""Per-stream timeline of the training step from a rocprofv3 kernel trace ("where do 46.9 - 39 ms go?").

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/step_loop.py --steps 6
    python tools/prof_timeline.py DIR/*/*_kernel_trace.csv OUT.md "title" [--steps 6]

Steps are cut at the BertAdam launch (`bertadam_kernel`: the last kernel of a step).  For every HIP stream (hardware queue)
of the last `--steps` steps: busy time (union of its kernels' intervals), idle time inside the step, and the idle time split by
cause -- gaps shorter than 4 us (back-to-back launch boundaries), gaps during which ANOTHER stream was running (the stream
waits on an event of that stream, or the host has not caught up because it is feeding the other stream), gaps in which
nothing ran on the GPU at all (host-bound / synchronisation).  The main stream's idle time is also listed by the kernel
family that FOLLOWS each gap, and the stretch of its kernels against an --alone trace (same step on one stream) is reported.
"""
import argparse
import collections
import csv
import re

FAMS = [("window conv fwd/dgrad", r"igemm_win_kernel"), ("ring / gather GEMM", r"igemm_kernel"),
        ("wgrad (sliding / temporal window)", r"wgrad_sw_kernel|wgrad_tw_kernel"), ("wgrad (generic, grouped)", r"wgrad_"),
        ("BatchNorm", r"bn_|partials"), ("attention", r"attention_"), ("LayerNorm", r"ln_"), ("conv0", r"conv0_"),
        ("elementwise", r"gelu|dropout|add_|transpose|colsum|softmax"), ("weight prep", r"prep_conv|unprep|cast_|select_taps|wn_|weightnorm"),
        ("heads + loss", r"sgemm|timepool|l2norm|cosnorm|loss_|hinge|spatial_mean|copy_f32|diag_|attnpool"),
        ("BertAdam", r"bertadam|sumsq|unscale"), ("input", r"video_normalize|collate|maxpool"), ("torch fill/copy", r"at::native|Memset|Memcpy|fill|rocclr")]


def fam(name):
    for label, pat in FAMS:
        if re.search(pat, name):
            return label
    return "other"


def union(iv):
    iv = sorted(iv)
    out, cur_s, cur_e = 0.0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                out += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        out += cur_e - cur_s
    return out


def overlap(s, e, others):
    """length of [s, e] covered by the sorted, disjoint intervals `others`"""
    tot = 0.0
    for a, b in others:
        if b <= s:
            continue
        if a >= e:
            break
        tot += min(e, b) - max(s, a)
    return tot


def merged(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("title")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, r["Queue_Id"] + "/" + r["Stream_Id"], r["Kernel_Name"]))
    rows.sort()
    ends = [e for s, e, q, n in rows if "bertadam_kernel" in n]
    if len(ends) < args.steps + 1:
        raise SystemExit(f"only {len(ends)} optimizer launches in the trace")
    cuts = ends[-(args.steps + 1):]
    streams = sorted({q for _, _, q, _ in rows})
    main_q = max(streams, key=lambda q: sum(e - s for s, e, qq, n in rows if qq == q and "igemm_win" in n))
    acc = {q: collections.defaultdict(float) for q in streams}
    follow = collections.defaultdict(float)
    span_tot = 0.0
    for k in range(args.steps):
        t0, t1 = cuts[k], cuts[k + 1]
        span_tot += t1 - t0
        step = [(s, e, q, n) for s, e, q, n in rows if s >= t0 and e <= t1]
        per_q = {q: [(s, e, n) for s, e, qq, n in step if qq == q] for q in streams}
        all_iv = merged([(s, e) for s, e, _, _ in step])
        for q in streams:
            ks = per_q[q]
            if not ks:
                continue
            others = merged([(s, e) for s, e, qq, _ in step if qq != q])
            a = acc[q]
            a["busy"] += union([(s, e) for s, e, _ in ks])
            a["kernels"] += len(ks)
            a["first"] += ks[0][0] - t0
            a["last"] += t1 - max(e for _, e, _ in ks)
            prev_e = t0
            for s, e, n in ks:
                gap = s - prev_e
                if gap > 0:
                    if gap < 4.0:
                        a["gap_launch"] += gap
                    else:
                        ov = overlap(prev_e, s, others)
                        a["gap_other_stream"] += ov
                        a["gap_gpu_idle"] += gap - ov
                    if q == main_q and gap >= 4.0:
                        follow[fam(n)] += gap
                prev_e = max(prev_e, e)
            a["gap_tail"] += t1 - prev_e
        acc["_all"] = acc.get("_all", collections.defaultdict(float))
        acc["_all"]["busy"] += sum(e - s for s, e in all_iv)
    n = args.steps
    with open(args.out, "w") as f:
        f.write(f"# {args.title}\n\nsource: `{args.trace}` (tools/prof_timeline.py), last {n} steps, cut at the BertAdam launch; "
                f"**{span_tot / n / 1e3:.2f} ms per step**, some kernel running on the GPU for {acc['_all']['busy'] / n / 1e3:.2f} ms of it.\n\n")
        f.write("| stream (queue/stream id) | kernels/step | busy ms | idle ms | of which: launch boundaries < 4 us | another stream "
                "running (waits on it / host feeding it) | nothing running (host / sync) | after its last kernel |\n|---|---|---|---|---|---|---|---|\n")
        for q in streams:
            a = acc[q]
            if not a["kernels"]:
                continue
            idle = span_tot - a["busy"]
            f.write(f"| {q}{' (video trunk)' if q == main_q else ''} | {a['kernels'] / n:.0f} | {a['busy'] / n / 1e3:.2f} | {idle / n / 1e3:.2f} | "
                    f"{a['gap_launch'] / n / 1e3:.2f} | {a['gap_other_stream'] / n / 1e3:.2f} | {a['gap_gpu_idle'] / n / 1e3:.2f} | {a['gap_tail'] / n / 1e3:.2f} |\n")
        f.write(f"\nIdle time of the trunk's stream ({main_q}) in gaps of 4 us or more, by the kernel family that follows the gap (ms per step):\n\n"
                "| next kernel | ms/step |\n|---|---|\n")
        for k, v in sorted(follow.items(), key=lambda kv: -kv[1]):
            f.write(f"| {k} | {v / n / 1e3:.2f} |\n")
    print(f"wrote {args.out}")


if __name__ == "__main__":
    main()
