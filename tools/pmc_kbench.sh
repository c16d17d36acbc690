#!/bin/bash
# Shader-core counters of single kernels under tools/kbench.py (four rocprofv3 --pmc passes).
# Usage (on the GPU box): [SCRIPT=tools/s1_bench.py] tools/pmc_kbench.sh <out-prefix> <script args...>
set -euo pipefail
OUT="${1:?usage: tools/pmc_kbench.sh <out-prefix> <args...>}"; shift
SCRIPT="${SCRIPT:-tools/kbench.py}"
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
PY="$(python -c 'import sys; print(sys.executable)')"  # (the program after -- must be the interpreter itself, not a shim)
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  # shellcheck disable=SC2086
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -- "$PY" "$SCRIPT" "$@" > "$OUT.g$i.log" 2>&1 \
    || { echo "group $i failed (see $OUT.g$i.log)" >&2; exit 1; }
done
"$PY" - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "anonymous" in k:
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    g = lambda c: d.get(c, float("nan"))
    mf = g("SQ_INSTS_VALU_MFMA_MOPS_F32") / 512.0  # MOPS counts 512 per 16x16x4 f32 MFMA? printed raw too
    print("%-40s n=%d" % (k[:40], max(len(v) for v in cs.values())))
    print("   mfma_busy %.3f  lds_util %.3f  lds_conflict/active %.3f  wait_inst_lds/wave %.3f  wait_inst_any/wave %.3f  wait_any/wave %.3f  active_any/wave %.3f  active_valu/wave %.3f" % (
        g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CU_CYCLES"),
        g("SQ_LDS_BANK_CONFLICT") / max(1, g("SQ_LDS_IDX_ACTIVE")), g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
        g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
    print("   insts: valu %.0f  lds %.0f  salu %.0f  mfma_mops %.0f  | wave_cycles %.0f busy_cu_cycles %.0f" % (
        g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SALU"), g("SQ_INSTS_VALU_MFMA_MOPS_F32"), g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CU_CYCLES")))
PY
