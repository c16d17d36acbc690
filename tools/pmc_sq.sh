#!/bin/bash
# Shader-core PMC counters per kernel (MFMA pipe busy, LDS bank conflicts, wait cycles), one rocprofv3 --pmc
# pass per counter group.  Usage: tools/pmc_sq.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT.g$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "curla" in k or "anonymous" in k:
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())} for k, cs in agg.items()}
json.dump(out, open("$OUT.json", "w"), indent=1)
def g(d, c): return d.get(c, float("nan"))
print("%-44s %6s %9s %9s %9s %9s %9s" % ("kernel", "n", "mfma_busy", "lds_confl", "wait_lds", "wait_any", "active"))
for k, d in sorted(out.items(), key=lambda kv: -g(kv[1], "GRBM_GUI_ACTIVE") * kv[1]["launches"])[:24]:
    busy = g(d, "SQ_BUSY_CU_CYCLES")
    print("%-44s %6d %9.3f %9.3f %9.3f %9.3f %9.3f" % (
        k[:44], d["launches"],
        g(d, "SQ_VALU_MFMA_BUSY_CYCLES") / busy if busy else 0,
        g(d, "SQ_LDS_BANK_CONFLICT") / max(1.0, g(d, "SQ_LDS_IDX_ACTIVE")),
        g(d, "SQ_WAIT_INST_LDS") / max(1.0, g(d, "SQ_WAVE_CYCLES")),
        g(d, "SQ_WAIT_INST_ANY") / max(1.0, g(d, "SQ_WAVE_CYCLES")),
        g(d, "SQ_ACTIVE_INST_ANY") / max(1.0, g(d, "SQ_WAVE_CYCLES"))))
PY
