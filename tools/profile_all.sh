#!/bin/bash
# Everything profiles/<tag>_* is made of, in one GPU call (about ten minutes of box time):
#   tools/profile_all.sh <tag>     then, back in the build container:  tools/profile_collect.sh <tag>
# per configuration (c2, c3, c5, c1t): bench line, rocprofv3 kernel stats, HBM traffic and shader-core counters
# (tools/profile_config.sh); the default bench command; the clock reconcile; the full-size parity reports.
TAG=${1:?tag}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out; mkdir -p $O
for CFG in c2 c3 c5 c1t; do
  echo "== $CFG"; tools/profile_config.sh $TAG $CFG 2>&1 | tail -4
done
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 300 $O/${TAG}_bench_default.json; echo
tools/clock_reconcile.sh $TAG > $O/${TAG}_clock.log 2>&1; tail -3 $O/${TAG}_clock_reconcile.txt
rm -f $O/fullsize_parity.txt $O/fullsize_arbiter.txt
python -m pytest tests/test_gpu_fullsize.py -q > $O/${TAG}_fullsize_tests.log 2>&1; tail -2 $O/${TAG}_fullsize_tests.log
cat $O/fullsize_parity.txt $O/fullsize_arbiter.txt > $O/${TAG}_fullsize_parity.txt
