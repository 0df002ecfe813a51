#!/bin/bash
# Developer probe: bench variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline"
run "share 4ctx" $B --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8
run "share 4ctx" $B --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8
run "half-file 4ctx (N=2 share)" $B --total-bytes 1073741824 --base-bytes 107374182 --steps 16 --warmup 4
run "full 4ctx" $B --steps 10 --warmup 3
