#!/bin/bash
# Developer probe: the walk's claim size and workgroups per XCD, side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_walk', k.get('k_walk'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 12 --warmup 4"
for cw in "1024 32" "1024 24" "2048 16" "4096 16" "1536 24" "768 48" "1024 32"; do set -- $cw; run "claim $1, walk wgs $2" env MI355X_BZ2_WALK_CHUNK=$1 MI355X_BZ2_WALK_WGS=$2 $B; done
run "claim 1024, wgs 32, walks side by side" env MI355X_BZ2_WALK_CHUNK=1024 MI355X_BZ2_WALK_WGS=32 MI355X_BZ2_WALK_SERIAL=0 $B
run "claim 1024, wgs 32, 5 contexts 16 queues" env MI355X_BZ2_WALK_CHUNK=1024 MI355X_BZ2_WALK_WGS=32 GPU_MAX_HW_QUEUES=16 $B --contexts 5
run "claim 1024, wgs 32, urandom" env MI355X_BZ2_WALK_CHUNK=1024 MI355X_BZ2_WALK_WGS=32 $B --workload urandom
run "default, urandom" $B --workload urandom
