#!/bin/bash
# Developer probe: decoder contexts x block groups x hardware queues, on ONE box, two rounds
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 6"
for rep in 1 2; do
run "4 contexts (default)" $B
run "4 contexts, no split" env MI355X_BZ2_NO_SPLIT=1 $B
run "5 contexts, no split, 16 queues" env GPU_MAX_HW_QUEUES=16 MI355X_BZ2_NO_SPLIT=1 $B --contexts 5
run "5 contexts, no split" env MI355X_BZ2_NO_SPLIT=1 $B --contexts 5
run "5 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $B --contexts 5
run "4 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $B
done
