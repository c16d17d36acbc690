#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of bench.py: per update, the kernel-busy time, the idle gaps between
consecutive kernels on the stream, and the kernels that follow the largest gaps.
Usage: tools/timeline_gaps.py <kernel_trace.csv> [n_last_updates] [--list]   (--list: the last two updates' launches in order)"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:48]


want_list = "--list" in sys.argv
if want_list:
    sys.argv.remove("--list")
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# an update starts at the gather of the sampled transitions' scalars (one launch per ReplayBuffer sample)
starts = [i for i, r in enumerate(rows) if "gather_transition_scalars" in r["Kernel_Name"] or "sample_stage" in r["Kernel_Name"]]
starts = starts[-(nlast + 1):]
tot_busy = tot_gap = 0.0
gap_by = collections.Counter()
n_by = collections.Counter()
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    wall = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    tot_busy += busy
    tot_gap += wall - busy
    for j in range(a + 1, b + 1):
        g = (int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3
        gap_by[short(rows[j]["Kernel_Name"])] += g
        n_by[short(rows[j]["Kernel_Name"])] += 1
n = len(starts) - 1
print(f"{n} updates: kernels busy {tot_busy / n:.1f} us, idle gaps {tot_gap / n:.1f} us per update, {sum(n_by.values()) / n:.0f} launches")
for k, g in gap_by.most_common(12):
    print(f"  gap before {k:50s} {g / n:7.1f} us/update over {n_by[k] / n:.1f} launches")
if want_list:
  for a, b in ((starts[-3], starts[-2]), (starts[-2], starts[-1])):  # two: the actor phase runs every other update
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"-- update of {b - a} launches")
    for j in range(a, b):
        r = rows[j]
        g = (int(r["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{j - a:3d} t={(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} gap {g:6.1f} dur {d:7.1f}  {short(r['Kernel_Name'])}")
