#!/bin/bash
# Everything the judged numbers of one configuration come from, in one GPU call:
#   bench line (HIP-event roofline + CPU baseline), rocprofv3 kernel stats of the same command, HBM traffic
#   (two --pmc passes: FETCH_SIZE, WRITE_SIZE) and shader-core counters (five --pmc passes).
# Usage (on the GPU box):  tools/profile_config.sh <tag> <c2|c3|c5> [steps] [warmup] [what]
#   what = any of: bench,stats,traffic,sq (default all).  Results land in gpurun_out/<tag>_*_<cfg>.*; copy the
#   summaries into profiles/ afterwards.  The PMC summaries record the hash of the kernel sources they were
#   measured on (bench.py quotes them only while that hash matches the built library).
set -e
TAG=${1:?tag}; CFG=${2:-c2}; STEPS=${3:-50}; WARM=${4:-10}; WHAT=${5:-bench,stats,traffic,sq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
# the program behind rocprofv3's `--` must be the interpreter itself, not a shim that execs it: with --pmc the profiler's
# preloaded library has initialised the GPU before the program starts, and an exec from such a process takes the box down
PY=$(python -c 'import sys; print(sys.executable)')
HASH=$($PY -c "from curla_amd import build; print(build.source_hash())")
PSTEPS=6; PWARM=2
if [ "$CFG" = "c5" ]; then PSTEPS=2; PWARM=1; fi
if [[ $WHAT == *bench* ]]; then
  $PY bench.py --config $CFG > $O/${TAG}_bench_${CFG}.json 2> $O/${TAG}_bench_${CFG}.err
  tail -1 $O/${TAG}_bench_${CFG}.json | cut -c1-400
fi
if [[ $WHAT == *stats* ]]; then
  rm -rf $O/prof_${TAG}_${CFG}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_${CFG} -- $PY bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline > $O/prof_${TAG}_${CFG}.log 2>&1
  F=$(find $O/prof_${TAG}_${CFG} -name "*kernel_stats.csv" | head -1)
  $PY tools/summarize_rocprof.py "$F" $O/${TAG}_kernel_stats_${CFG}.txt "bench.py --config $CFG --steps $STEPS --warmup $WARM (sources $HASH)"
  head -14 $O/${TAG}_kernel_stats_${CFG}.txt
  # raw traces are large (gpurun copies back at most 64 MiB): keep the per-dispatch trace only when asked to
  if [ -n "$KEEP_TRACE" ]; then cp "$(find $O/prof_${TAG}_${CFG} -name "*kernel_trace.csv" | head -1)" $O/${TAG}_kernel_trace_${CFG}.csv; fi
  rm -rf $O/prof_${TAG}_${CFG}
fi
if [[ $WHAT == *traffic* ]]; then
  P=$O/pmc_traffic_${TAG}_${CFG}; rm -rf $P
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- $PY bench.py --config $CFG --steps $PSTEPS --warmup $PWARM --no-cpu-baseline --clock-warmup-s 0 > $P.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- $PY bench.py --config $CFG --steps $PSTEPS --warmup $PWARM --no-cpu-baseline --clock-warmup-s 0 > $P.write.log 2>&1
  $PY tools/pmc_summarize.py traffic $P $O/${TAG}_pmc_traffic_${CFG}.json $HASH | head -10
  rm -rf $P
fi
if [[ $WHAT == *sq* ]]; then
  P=$O/pmc_sq_${TAG}_${CFG}; rm -rf $P
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_LDS SQ_CYCLES" \
             "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $P/g$i -- $PY bench.py --config $CFG --steps $PSTEPS --warmup $PWARM --no-cpu-baseline --clock-warmup-s 0 > $P.g$i.log 2>&1
  done
  $PY tools/pmc_summarize.py sq $P $O/${TAG}_pmc_sq_${CFG}.json $HASH | tee $O/${TAG}_pmc_sq_${CFG}.txt | head -14
  rm -rf $P
fi
