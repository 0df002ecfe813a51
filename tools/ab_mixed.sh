#!/bin/bash
# Developer probe: the expensive group's scan with one wave per block (0) or with k_hscan_spec<4 / 8>, on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'k_hscan', d['roofline']['kernels_ms']['k_hscan'])"; }
B="python bench.py --no-cpu-baseline --steps 16 --warmup 4"
O="python bench.py --no-cpu-baseline --steps 6 --warmup 2 --contexts 1 --resident"
for v in 0 1 4 8; do
run "four contexts, mixed=$v" env MI355X_BZ2_SCAN_MIXED=$v $B
done
for v in 0 1 4 8; do
run "one context resident, mixed=$v" env MI355X_BZ2_SCAN_MIXED=$v $O
done
