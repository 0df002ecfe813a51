#!/bin/bash
# Developer probe: library variants (build/variants/*.so) side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_walk', k.get('k_walk'), 'k_mtf', k.get('k_mtf<272>'), k.get('k_mtf<144>'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 4"
for lib in "$@"; do
run "$lib" env MI355X_BZ2_LIBRARY=$PWD/build/variants/$lib.so MI355X_BZ2_WALK_CHUNK=256 $B
done
run "tree" $B
