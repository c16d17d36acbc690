#!/bin/bash
# copy what tools/profile_all.sh <tag> left in gpurun_out/ into profiles/ (tracked)
TAG=${1:?tag}
cd "$(dirname "$0")/.."
for CFG in c2 c3 c5 c1t; do
  for F in bench_$CFG.json kernel_stats_$CFG.txt pmc_traffic_$CFG.json pmc_sq_$CFG.json pmc_sq_$CFG.txt; do
    [ -f gpurun_out/${TAG}_$F ] && cp gpurun_out/${TAG}_$F profiles/
  done
done
for F in bench_default.json clock_reconcile.txt clock_reconcile.json fullsize_parity.txt; do
  [ -f gpurun_out/${TAG}_$F ] && cp gpurun_out/${TAG}_$F profiles/
done
ls -la profiles/${TAG}_* | wc -l
