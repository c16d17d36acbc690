#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs into per-kernel means per launch.

    pmc_summarize.py traffic <dir with fetch/ and write/> <out.json> <source-hash>
    pmc_summarize.py sq      <dir with g1..g4/>           <out.json> <source-hash>

traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B/lane stores.
The output records the hash of the kernel sources the counters were measured on ("_source_hash")."""
import collections
import csv
import glob
import json
import sys


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def ours(k):
    return "curla" in k or "anonymous" in k


def per_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(kind, d, out, src_hash):
    res = {"_source_hash": src_hash}
    if kind == "traffic":
        fe, wr = per_kernel(f"{d}/fetch/*/*counter_collection.csv"), per_kernel(f"{d}/write/*/*counter_collection.csv")
        for k, c in fe.items():
            f_kib = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            w = wr.get(k, {}).get("WRITE_SIZE", [0.0])
            w_kib = sum(w) / len(w)
            res[k] = {"launches": len(c["FETCH_SIZE"]), "fetch_raw_bytes": f_kib * 1024,
                      "fetch_corrected_bytes": 2 * f_kib * 1024, "write_bytes": w_kib * 1024,
                      "traffic_bytes": 2 * f_kib * 1024 + w_kib * 1024}
        json.dump(res, open(out, "w"), indent=1)
        rows = [(k, v) for k, v in res.items() if not k.startswith("_")]
        for k, v in sorted(rows, key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:14]:
            print("%-44s n=%4d fetch(x2) %8.1f MB  write %8.1f MB" % (k[:44], v["launches"], v["fetch_corrected_bytes"] / 1e6,
                                                                        v["write_bytes"] / 1e6))
    else:
        agg = per_kernel(f"{d}/g*/*/*counter_collection.csv")
        for k, cs in agg.items():
            res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
            res[k]["launches"] = max(len(v) for v in cs.values())
        json.dump(res, open(out, "w"), indent=1)

        def g(dd, c):
            return dd.get(c, float("nan"))
        print("# rocprofv3 --pmc (4 separate passes); means per launch; kernel sources %s" % src_hash)
        print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); lds_util = SQ_LDS_IDX_ACTIVE / "
              "SQ_BUSY_CU_CYCLES; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait_any = SQ_WAIT_INST_ANY / "
              "SQ_WAVE_CYCLES; valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_VALU_MFMA_MOPS_F32-derived MFMA count")
        print("%-46s %5s %9s %9s %12s %9s" % ("kernel", "n", "mfma_busy", "lds_util", "lds_conflict", "wait_any"))
        rows = [(k, v) for k, v in res.items() if not k.startswith("_")]
        for k, dd in sorted(rows, key=lambda kv: -g(kv[1], "GRBM_GUI_ACTIVE") * kv[1]["launches"])[:30]:
            busy = g(dd, "SQ_BUSY_CU_CYCLES")
            print("%-46s %5d %9.3f %9.3f %12.3f %9.3f" % (
                k[:46], dd["launches"], g(dd, "SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * busy) if busy else 0,
                g(dd, "SQ_LDS_IDX_ACTIVE") / busy if busy else 0,
                g(dd, "SQ_LDS_BANK_CONFLICT") / max(1.0, g(dd, "SQ_LDS_IDX_ACTIVE")),
                g(dd, "SQ_WAIT_INST_ANY") / max(1.0, g(dd, "SQ_WAVE_CYCLES"))))


if __name__ == "__main__":
    main(*sys.argv[1:5])
