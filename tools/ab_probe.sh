#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
S="--total-bytes 268435456 --base-bytes 26843545 --steps 60 --warmup 8 --no-cpu-baseline"
for q in 16 24 32 16 24; do run "share 4ctx hw queues $q" env GPU_MAX_HW_QUEUES=$q python bench.py $S; done
run "full 4ctx q16" env GPU_MAX_HW_QUEUES=16 python bench.py --no-cpu-baseline --steps 16 --warmup 4
run "full 4ctx q24" env GPU_MAX_HW_QUEUES=24 python bench.py --no-cpu-baseline --steps 16 --warmup 4
