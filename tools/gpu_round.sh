#!/bin/bash
# One GPU-box round: tests, smoke, bench, rocprof kernel stats.  Usage: tools/gpu_round.sh <tag> [bench args]
set -o pipefail
TAG=${1:-r}
shift
mkdir -p gpurun_out
R=$PWD
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest_exit=$?"; tail -5 gpurun_out/pytest_$TAG.log
echo "== smoke"
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke_exit=$?"; tail -3 gpurun_out/smoke_$TAG.log
echo "== bench"
timeout -k 10 900 python bench.py "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench_exit=$?"; tail -3 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
echo "== rocprofv3 kernel stats"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1; echo "rocprof_exit=$?"
cd $R
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
