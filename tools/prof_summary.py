"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a per-step table (markdown)."""
import csv
import sys


def main(path, steps, out, title):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `{path}` ({steps} steps profiled); total kernel time {tot / steps / 1e6:.2f} ms/step\n\n")
        f.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows:
            if float(r["Percentage"]) < 0.05:
                continue
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0][:90]
            f.write(f"| `{name}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4])
