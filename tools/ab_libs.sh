#!/bin/bash
# Developer probe: library variants (build/variants/*.so) side by side on ONE box.  usage: ab_libs.sh <bench args> -- lib...
ARGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ARGS+=("$1"); shift; done
shift
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_hscan', k.get('k_hscan'), 'k_walk', k.get('k_walk'), 'k_mtf', k.get('k_mtf<272>'), k.get('k_mtf<144>'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 12 --warmup 4 ${ARGS[@]}"
for lib in "$@"; do
run "$lib" env MI355X_BZ2_LIBRARY=$PWD/build/variants/$lib.so $B
done
run "tree" $B
