#!/bin/bash
# Texture-addresser / L1 counters of single kernels under tools/kbench.py (rocprofv3 --pmc passes).
# Usage (on the GPU box): tools/pmc_ta.sh <out-prefix> <kbench args...>
set -euo pipefail
OUT="${1:?usage: tools/pmc_ta.sh <out-prefix> <kbench args...>}"; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
# the program after `--` must be the interpreter itself: with --pmc the profiler has initialised the GPU before the
# program starts, and a shim script that re-execs would be an exec inside a GPU-initialised process
PY="$(python -c 'import sys; print(sys.executable)')"
i=0
failed=0
# (only the busy counters: a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum /
# TA_BUFFER_WAVEFRONTS_sum aborted rocprofv3 on this pool and left the run hanging until gpurun's silence limit)
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  # shellcheck disable=SC2086  # ($grp is a list of counter names)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/t$i" -- "$PY" tools/kbench.py "$@" > "$OUT.t$i.log" 2>&1 \
    || { echo "group $i failed (see $OUT.t$i.log)" >&2; failed=1; }
done
[ "$failed" -eq 0 ] || exit 1
"$PY" - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/t*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "anonymous" in k:
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k[:50])
    for c, v in sorted(cs.items()):
        print("   %-44s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
