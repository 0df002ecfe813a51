#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
S="--total-bytes 268435456 --base-bytes 26843545 --steps 60 --warmup 8 --no-cpu-baseline"
for q in 8 12 16 20; do run "share 4ctx hw queues $q" env GPU_MAX_HW_QUEUES=$q python bench.py $S; done
for q in 12 16 20; do run "full 4ctx hw queues $q" env GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 16 --warmup 4; done
run "share 3ctx hw queues 12" env GPU_MAX_HW_QUEUES=12 python bench.py $S --contexts 3
run "share 3ctx hw queues 16" env GPU_MAX_HW_QUEUES=16 python bench.py $S --contexts 3
