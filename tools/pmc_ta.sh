#!/bin/bash
# Texture-addresser / L1 counters of single kernels under tools/kbench.py (rocprofv3 --pmc passes).
# Usage (on the GPU box): tools/pmc_ta.sh <out-prefix> <kbench args...>
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
# (only the busy counters: a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum /
# TA_BUFFER_WAVEFRONTS_sum aborted rocprofv3 on this pool and left the run hanging until gpurun's silence limit)
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/t$i -- python tools/kbench.py "$@" > $OUT.t$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
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
