#!/bin/bash
# Whole-update throughput of one configuration under each value of each run-time option (one at a time, the others at
# their defaults): tools/option_sweep.sh <c2|c3|c5> [steps]   -> gpurun_out/option_sweep_<cfg>.txt
CFG=${1:-c2}; STEPS=${2:-80}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/option_sweep_$CFG.txt; : > $O
run() {  # label, env assignments...
  local label=$1; shift
  local v=$(env "$@" python bench.py --config $CFG --steps $STEPS --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f update()/s  %.3f ms' % (d['value'], d['ms_per_step']))")
  printf "%-28s %s\n" "$label" "$v" | tee -a $O
}
run "defaults" X=1
run "defaults (again)" X=1
for v in band rw; do run "conv1_u8=$v" CURLA_CONV1_U8=$v; done
for v in 0 1; do run "bwd_split=$v" CURLA_BWD_SPLIT=$v; done
for v in f23 f43; do run "s1_fwd=$v" CURLA_S1_FWD=$v; done
for v in 6464 6432 3232 12864; do run "gemm_tile=$v" CURLA_GEMM_TILE=$v; done
run "gemm_mfma=b3" CURLA_GEMM_MFMA=b3
run "gemm_mfma=f32" CURLA_GEMM_MFMA=f32
run "s1_wgrad=x" CURLA_S1_WGRAD=x
run "wgrad1_u8=f32" CURLA_WGRAD1_U8=f32
run "linear_bwd=split" CURLA_LINEAR_BWD=split
run "CURLA_FC_FWD=gemm" CURLA_FC_FWD=gemm
run "CURLA_CURL_HEAD=unfused" CURLA_CURL_HEAD=unfused
run "CURLA_TORCH_ADAM=1" CURLA_TORCH_ADAM=1
