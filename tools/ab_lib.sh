#!/bin/bash
# Developer probe: two builds of the library side by side on ONE box (MI355X_BZ2_LIBRARY), runs interleaved
# usage: tools/ab_lib.sh <other library in indexed_bzip2_amd/> [rounds]
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: walk', k.get('k_walk'), 'emit', k.get('k_emit'), 'bwt', k.get('k_bwt_build'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 6"
for rep in $(seq 1 ${2:-3}); do
run "this build" $B
run "$1" env MI355X_BZ2_LIBRARY=$PWD/indexed_bzip2_amd/$1 $B
done
run "this build, urandom" $B --workload urandom
run "$1, urandom" env MI355X_BZ2_LIBRARY=$PWD/indexed_bzip2_amd/$1 $B --workload urandom
