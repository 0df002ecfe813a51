#!/bin/bash
# Developer probe: bench variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
run "full 4ctx" $B
run "full 4ctx" $B
run "full 4ctx resident" $B --resident
run "share 4ctx" $B --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8
