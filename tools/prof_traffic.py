"""HBM traffic per matrix-core launch, by shape, from two rocprofv3 PMC passes over tools/step_loop.py --isolated
(FETCH_SIZE and WRITE_SIZE in separate passes: together they exceed the four TCC slots), joined with the launch log like
tools/prof_shapes.py.  FETCH_SIZE is KiB and counts wide streaming reads at HALF on gfx950 (MI355X_MICROARCH.md, HBM
section): doubled here; WRITE_SIZE (KiB) is taken as read.

    python tools/prof_traffic.py FETCH_DIR WRITE_DIR LAUNCH.json OUT.md OUT.json
OUT.json maps bench.py's roofline kernel key ("family<mode> N=.. K=..", the largest-M shape of that key) to bytes per launch:
bench.py's `roofline.traffic` reads it."""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from prof_shapes import join, flops, PEAK


def counters(d):
    vals = collections.defaultdict(float)
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            vals[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    return vals


def main(fetch_dir, write_dir, log, out_md, out_json):
    ft = glob.glob(f"{fetch_dir}/**/*kernel_trace.csv", recursive=True)[0]
    wt = glob.glob(f"{write_dir}/**/*kernel_trace.csv", recursive=True)[0]
    tf, meta = join(ft, log)
    tw, _ = join(wt, log)
    cf, cw = counters(fetch_dir), counters(write_dir)
    rows = []
    for key, rec in tf.items():
        rd = sum(cf.get((i, "FETCH_SIZE"), 0.0) for i in rec["ids"]) / len(rec["ids"]) * 1024 * 2
        wr = sum(cw.get((i, "WRITE_SIZE"), 0.0) for i in tw[key]["ids"]) / len(tw[key]["ids"]) * 1024 if key in tw else float("nan")
        name, mode, M, N, K, nb, taps, stride = key
        ntap = taps[0] * taps[1] * taps[2]
        if name.startswith("wgrad"):          # X rows (K / taps channels each) + dY rows + dW
            alg = (M * (K // ntap) + M * N) * 2.0 * nb + N * K * 4.0 * nb
        else:                                 # A rows + weights + C rows (stride-1 taps re-use rows: compulsory = once)
            a_cols = K // ntap if mode != "dense" else K
            sdiv = stride[0] * stride[1] * stride[2] if mode == "conv_dgrad" else 1
            smul = stride[0] * stride[1] * stride[2] if mode == "conv_fwd" else 1
            alg = (M * smul / sdiv * a_cols + N * K + M * N) * 2.0 * nb
        us = sum(rec["us"]) / len(rec["us"])
        rows.append((len(rec["us"]) / meta["steps"] * us, key, rd, wr, alg, us))
    rows.sort(reverse=True)
    table = {}
    with open(out_md, "w") as f:
        f.write("# HBM bytes per matrix-core launch by shape (PMC)\n\n"
                f"`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` over `tools/step_loop.py --isolated` "
                f"({meta['steps']} steps, batch {meta['batch']}), joined with the launch log (tools/prof_traffic.py).  read = FETCH_SIZE KiB x 2 "
                "(the gfx950 correction), written = WRITE_SIZE KiB.  algorithmic = operand rows once + weights + output (no halo, no "
                "statistics rows); ratio = (read + written) / algorithmic.  Under the profiler kernels run one at a time.\n\n"
                "| kernel | mode | M | N | K | taps | stride | avg us (profiled) | read MB | written MB | algorithmic MB | ratio | GB/s |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for tot, key, rd, wr, alg, us in rows[:60]:
            name, mode, M, N, K, nb, taps, stride = key
            f.write(f"| `{name}` | {mode} | {M}{' x' + str(nb) if nb > 1 else ''} | {N} | {K} | {'x'.join(map(str, taps))} | {'x'.join(map(str, stride))} | "
                    f"{us:.1f} | {rd / 1e6:.0f} | {wr / 1e6:.0f} | {alg / 1e6:.0f} | {(rd + wr) / alg:.2f} | {(rd + wr) / us / 1e3:.0f} |\n")
            fam = name.split("<")[0]
            bkey = f"{fam}<{mode}> N={N} K={K}" if not fam.startswith("wgrad") else f"{fam}<{mode}> Ni={N} Kj={K}"
            if bkey not in table or M > table[bkey][0]:
                table[bkey] = (M, rd + wr)
    json.dump({k: v[1] for k, v in table.items()}, open(out_json, "w"), indent=1)
    print(f"wrote {out_md}, {out_json}")


if __name__ == "__main__":
    main(*sys.argv[1:6])
