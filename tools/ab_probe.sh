#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'k_hscan', d['roofline']['kernels_ms']['k_hscan'])"; }
B="python bench.py --no-cpu-baseline --steps 16 --warmup 4"
N="env MI355X_BZ2_NO_SPLIT=1 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --contexts 1 --resident"
run "nosplit margin 12%" $N
run "nosplit margin 6%" env MI355X_BZ2_SCAN_TUNE=4 $N
run "nosplit margin 3%" env MI355X_BZ2_SCAN_TUNE=8 $N
run "full margin 12%" $B
run "full margin 6%" env MI355X_BZ2_SCAN_TUNE=4 $B
run "full margin 3%" env MI355X_BZ2_SCAN_TUNE=8 $B
