#!/bin/bash
# HBM traffic of the conv kernels from rocprofv3 PMC counters (two separate passes: FETCH_SIZE and WRITE_SIZE
# do not fit one pass on gfx950).  Usage: tools/pmc_traffic.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT.write.log 2>&1
python - <<PY
import csv, glob, collections, json
def per_kernel(d):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg
fe, wr = per_kernel("fetch"), per_kernel("write")
out = {}
for k in fe:
    if "curla" in k or "anonymous" in k:
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
        # coalesced stream -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B/lane stores
        f_kib = sum(fe[k]) / len(fe[k]); w_kib = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
        out[short] = {"launches": len(fe[k]), "fetch_raw_bytes": f_kib * 1024, "fetch_corrected_bytes": 2 * f_kib * 1024,
                      "write_bytes": w_kib * 1024, "traffic_bytes": 2 * f_kib * 1024 + w_kib * 1024}
json.dump(out, open("$OUT.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:12]:
    print("%-40s n=%4d fetch(x2) %8.1f MB  write %8.1f MB" % (k[:40], v["launches"], v["fetch_corrected_bytes"] / 1e6, v["write_bytes"] / 1e6))
PY
