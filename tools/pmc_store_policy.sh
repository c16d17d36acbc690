#!/bin/bash
# WRITE_SIZE of tools/micro/store_policy.hip's four kernels (one rocprofv3 --pmc pass).  On the GPU box:
#   tools/pmc_store_policy.sh gpurun_out/store_policy
set -euo pipefail
OUT="${1:?usage: tools/pmc_store_policy.sh <out-prefix>}"
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run this on the GPU box}"
BIN=tools/_build/store_policy
[ -x "$BIN" ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -o "$BIN" tools/micro/store_policy.hip
"./$BIN" > "$OUT.times.txt"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT" -- "./$BIN" > "$OUT.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
stored = (4 << 20) * 128
print("# WRITE_SIZE (KB units per MI355X_MICROARCH.md: x 1024 bytes) per launch against the 512 MiB each kernel stores")
for k, v in agg.items():
    b = sum(v) / len(v) * 1024
    print(f"{k:24s} WRITE_SIZE {b / 1e6:9.1f} MB  = {b / stored:5.2f} x the bytes stored")
PY
cat "$OUT.times.txt"
