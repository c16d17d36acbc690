#!/usr/bin/env python3
"""tools/clock_reconcile.sh's last step: the in-kernel clock / utilisation (tools/clock_reconcile.py's JSON) beside the
PMC counters of the SAME launches (the last N dispatches of conv_rwb_fwd_kernel in each rocprofv3 pass).
    clock_summarize.py <gpurun_out> <out.txt>"""
import csv
import glob
import json
import sys

O, out_path = sys.argv[1], sys.argv[2]
src_hash = sys.argv[3] if len(sys.argv) > 3 else None
summary = {"_source_hash": src_hash, "kernel": "conv_rwb_fwd_kernel"}
lines = []
say = lines.append


def last_rows(pattern, n):
    per = {}
    for f in glob.glob(pattern):
        rows = [r for r in csv.DictReader(open(f)) if "conv_rwb_fwd_kernel" in r["Kernel_Name"]]
        for r in rows:
            per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {c: [v for _, v in sorted(vs)][-n:] for c, vs in per.items()}


def durations(pattern, n):
    d = []
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if "conv_rwb_fwd_kernel" in r["Kernel_Name"]:
                d.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    return [v for _, v in sorted(d)][-n:]


mean = lambda v: sum(v) / max(1, len(v))  # noqa: E731
say("# conv_rwb_fwd_kernel: the in-kernel clock, the matrix-pipe utilisation and the PMC counters of the SAME launches")
say("# (tools/clock_reconcile.sh; diagnostic build -DRWB_CLOCK: two stamps per workgroup and launch, nothing in the loops)")
for mode in ("stack", "update"):
    try:
        plain = json.load(open(f"{O}/clock_{mode}_plain.json"))
    except OSError:
        continue
    say("")
    say(f"== {plain['what']}")
    summary[mode] = {"what": plain["what"], "in_kernel_clock_GHz": plain["in_kernel_clock_GHz"]["median"],
                     "kernel_us": plain["kernel_us_in_kernel"]["max"],
                     "matrix_pipe_util_at_in_kernel_clock": plain["matrix_pipe_util_at_in_kernel_clock"],
                     "issued_TFLOPs": plain["issued_TFLOPs"], "mfma_per_launch": plain["mfma_per_launch_from_shapes"]}
    for tag, fn in (("un-profiled", f"clock_{mode}_plain.json"), ("under rocprofv3 --pmc, pass 1", f"clock_{mode}_pmc1.json"),
                    ("under rocprofv3 --pmc, pass 2", f"clock_{mode}_pmc2.json")):
        try:
            r = json.load(open(f"{O}/{fn}"))
        except OSError:
            continue
        say(f"  [{tag}] in-kernel clock {r['in_kernel_clock_GHz']['median']:.3f} GHz (workgroups {r['in_kernel_clock_GHz']['min']:.3f}"
            f" .. {r['in_kernel_clock_GHz']['max']:.3f}); {r['shader_cycles_per_launch']['median']:.0f} shader cycles and "
            f"{r['kernel_us_in_kernel']['median']:.1f} us per launch (slowest workgroup {r['shader_cycles_per_launch']['max']:.0f} / "
            f"{r['kernel_us_in_kernel']['max']:.1f}); {r['mfma_per_launch_from_shapes'] / 1e6:.2f} M matrix instructions per launch")
        say(f"      -> matrix pipe busy {r['matrix_pipe_util_at_in_kernel_clock']:.3f} of the SIMD cycles at that clock; "
            f"{r['issued_TFLOPs']:.0f} TFLOP/s issued = {r['frac_of_2500_TF_nominal_peak']:.3f} of the 2.4 GHz nominal bf16 peak")
    n = json.load(open(f"{O}/clock_{mode}_pmc1.json"))["launches"] if glob.glob(f"{O}/clock_{mode}_pmc1.json") else 0
    if not n:
        continue
    c1 = last_rows(f"{O}/clock_pmc_{mode}/g1/*/*counter_collection.csv", n)
    c2 = last_rows(f"{O}/clock_pmc_{mode}/g2/*/*counter_collection.csv", n)
    us1 = durations(f"{O}/clock_pmc_{mode}/g1/*/*kernel_trace.csv", n)
    m = {k: mean(v) for k, v in {**c1, **c2}.items()}
    r1 = json.load(open(f"{O}/clock_{mode}_pmc1.json"))
    cyc = r1["shader_cycles_per_launch"]["max"]
    say(f"  PMC, means over the same {n} launches (pass 1; kernel-trace duration {mean(us1):.1f} us):")
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_CYCLES",
              "SQ_BUSY_CYCLES"):
        if k in m:
            say(f"      {k:28s} {m[k]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        busy, cu = m["SQ_VALU_MFMA_BUSY_CYCLES"], m.get("SQ_BUSY_CU_CYCLES", 0.0)
        say(f"      SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)            = {busy / (4 * cu) if cu else 0:.3f}   (bench.py's mfma_busy_frac_pmc)")
        summary[mode]["pmc"] = {k: m[k] for k in m}
        summary[mode]["pmc_in_kernel_clock_GHz"] = r1["in_kernel_clock_GHz"]["median"]
        say(f"      SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x in-kernel cycles)    = {busy / (1024 * cyc):.3f}   (same launches, stamps)")
        say(f"      SQ_BUSY_CU_CYCLES / 256 CUs                                   = {cu / 256:.0f} against {cyc:.0f} in-kernel shader cycles: ratio {cu / 256 / cyc:.3f}")
        if "SQ_WAVE_CYCLES" in m:
            say(f"      SQ_WAVE_CYCLES x 4 / 2048 waves                               = {m['SQ_WAVE_CYCLES'] * 4 / 2048:.0f}: ratio {m['SQ_WAVE_CYCLES'] * 4 / 2048 / cyc:.3f}")
        if "GRBM_GUI_ACTIVE" in m and us1:
            say(f"      GRBM_GUI_ACTIVE / 8 XCDs / kernel-trace duration              = {m['GRBM_GUI_ACTIVE'] / 8 / mean(us1) / 1e3:.3f} GHz against the in-kernel {r1['in_kernel_clock_GHz']['median']:.3f} GHz")
    if "SQ_INSTS_MFMA" in m:
        say(f"      SQ_INSTS_MFMA against the count from the shapes               = {m['SQ_INSTS_MFMA'] / r1['mfma_per_launch_from_shapes']:.4f}")
open(out_path, "w").write("\n".join(lines) + "\n")
with open(out_path.rsplit(".", 1)[0] + ".json", "w") as f:
    json.dump(summary, f, indent=1)
