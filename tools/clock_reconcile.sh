#!/bin/bash
# One number for the dominant kernel's pipe utilisation, with its clock: the in-kernel clock
# (s_memtime / s_memrealtime), MFMA count x 16 / (1024 SIMDs x elapsed shader cycles) and the PMC counters
# SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_MFMA of the SAME launches
# (the diagnostic build runs under rocprofv3 --pmc), next to the un-profiled run of the same script.
#   tools/clock_reconcile.sh [tag]    ->  gpurun_out/<tag>_clock_reconcile.txt
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PY=$(python -c 'import sys; print(sys.executable)')
O=gpurun_out; mkdir -p $O
# (rebuilt when any kernel source is newer than it: the summary records the hash of the sources in the tree)
NEWEST=$(ls -t curla_amd/csrc/*.hip curla_amd/csrc/*.h | head -1)
if [ ! -f tools/_build/libcurla_clock.so ] || [ "$NEWEST" -nt tools/_build/libcurla_clock.so ]; then
  tools/build_variant.sh clock -DRWB_CLOCK > /dev/null
fi
export CURLA_LIB_PATH=$PWD/tools/_build/libcurla_clock.so
for MODE in stack update; do
  $PY tools/clock_reconcile.py --mode $MODE --out $O/clock_${MODE}_plain.json > /dev/null
  P=$O/clock_pmc_$MODE; rm -rf $P
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/g1 -- $PY tools/clock_reconcile.py --mode $MODE --settle-s 1.0 --launches 20 --out $O/clock_${MODE}_pmc1.json > $P.g1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $P/g2 -- $PY tools/clock_reconcile.py --mode $MODE --settle-s 1.0 --launches 20 --out $O/clock_${MODE}_pmc2.json > $P.g2.log 2>&1
done
HASH=$($PY -c "from curla_amd import build; print(build.source_hash())")
$PY tools/clock_summarize.py $O $O/${TAG}_clock_reconcile.txt $HASH
cat $O/${TAG}_clock_reconcile.txt
rm -rf $O/clock_pmc_stack $O/clock_pmc_update
