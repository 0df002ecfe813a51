#!/bin/bash
# Developer probe: more decoder contexts with fewer streams each, 16 hardware queues, on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 6"
run "4 contexts (default)" $B
for c in 5 6 8; do
run "$c contexts, no split, 16 queues" env GPU_MAX_HW_QUEUES=16 MI355X_BZ2_NO_SPLIT=1 $B --contexts $c
run "$c contexts, one chunk + expensive, 16 queues" env GPU_MAX_HW_QUEUES=16 MI355X_BZ2_CHUNKS=1 $B --contexts $c
done
run "4 contexts, no split, 16 queues" env GPU_MAX_HW_QUEUES=16 MI355X_BZ2_NO_SPLIT=1 $B
run "3 contexts" $B --contexts 3
