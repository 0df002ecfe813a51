#!/bin/bash
# Developer probe: the walk's claim size and workgroups per XCD, side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_walk', k.get('k_walk'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 12 --warmup 4"
for cw in ${AB_WALK:-"1024 32" "256 128" "512 64" "256 64" "2048 32"}; do set -- $cw; run "claim $1, walk wgs $2" env MI355X_BZ2_WALK_CHUNK=$1 MI355X_BZ2_WALK_WGS=$2 $B; done
