#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short
table (kernel names shortened) for profiles/."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*?>)?)\(", name)
    if m:
        name = m.group(1)
    if name.startswith("at::native"):
        for key in ("FusedAdamMathFunctor<double", "FusedAdamMathFunctor<float", "random_from_to", "normal_kernel",
                    "uniform_kernel", "FillFunctor", "scatter_gather", "BinaryOpScalarFunctor"):
            if key in name:
                return "torch:" + key
        return "torch:" + name[:60]
    return name[:90]


def main(path, out, what=""):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python {what}\n" if what else "")
        f.write(f"# source: {path}\n# total kernel time {tot / 1e6:.3f} ms\n")
        f.write(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
        for r in rows:
            f.write(f"{short(r['Name']):70s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                    f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['Percentage']):7.2f}\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
