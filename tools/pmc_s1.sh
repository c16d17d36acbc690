#!/bin/bash
# Shader-core counters of the stride-1 forward stack under one value of option s1_fwd (tools/s1_bench.py):
#   tools/pmc_s1.sh <impl> [--c5]      -> gpurun_out/pmc_s1_<impl>.txt
# matrix-pipe busy share, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) and VALU / MFMA instruction counts.
set -e
IMPL=${1:?impl}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(python -c 'import sys; print(sys.executable)')
O=gpurun_out/pmc_s1_$IMPL; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- $PY tools/s1_bench.py --impls $IMPL "$@" > $O/g$i.log 2>&1 || { tail -5 $O/g$i.log; exit 1; }
done
$PY - "$O" <<'PY' | tee gpurun_out/pmc_s1_$IMPL.txt
import collections, csv, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_rw" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/g1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_rw" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    us = sum(dur[k]) / max(1, len(dur[k]))
    busy = m.get("SQ_BUSY_CU_CYCLES", 0)
    print(k.split("(")[0])
    print("  launches %d  avg %.1f us (under the profiler)" % (len(dur[k]), us))
    print("  matrix pipe busy %.3f of CU-busy cycles; effective clock %.2f GHz" % (
        m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * busy) if busy else 0, m.get("GRBM_GUI_ACTIVE", 0) / 8 / us / 1e3))
    print("  per launch: MFMA %.1f M  VALU %.1f M (%.2f per MFMA)  LDS %.1f M;  wait_any %.3f of wave cycles" % (
        m.get("SQ_INSTS_MFMA", 0) / 1e6, m.get("SQ_INSTS_VALU", 0) / 1e6,
        m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_INSTS_MFMA", 1)), m.get("SQ_INSTS_LDS", 0) / 1e6,
        m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))))
    wc = max(1.0, m.get("SQ_WAVE_CYCLES", 1))
    print("  shares of wave cycles: " + "  ".join("%s %.3f" % (n.replace("SQ_", ""), m[n] / wc) for n in sorted(m)
                                                    if n.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES"))))
    print("  raw means per launch: " + "  ".join("%s %.3g" % (n.replace("SQ_", ""), m[n]) for n in sorted(m)))
PY
rm -rf $O
