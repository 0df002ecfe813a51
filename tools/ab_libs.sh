#!/bin/bash
# Developer probe: library variants (build/variants/*.so) side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_hscan', k.get('k_hscan'), 'k_mtf', k.get('k_mtf<272>'), k.get('k_mtf<144>'), 'k_link2', k.get('k_link2'), 'k_hsym', k.get('k_hsym'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 4"
for lib in "$@"; do
run "$lib" env MI355X_BZ2_LIBRARY=$PWD/build/variants/$lib.so $B
done
run "tree" $B
