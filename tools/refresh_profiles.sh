set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err || true
tail -1 gpurun_out/bench_n1.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1e -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof_r1e.log 2>&1 || true
F=$(find gpurun_out/prof_r1e -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py "$F" gpurun_out/r01_e_kernel_stats.txt
head -12 gpurun_out/r01_e_kernel_stats.txt
bash tools/pmc_traffic.sh gpurun_out/pmc_traffic 2>&1 | tail -8
