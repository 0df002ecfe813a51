#!/bin/bash
# Developer probe: config 3 (incompressible data) by contexts and block groups, on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --workload urandom --no-cpu-baseline --no-host-output --steps 10 --warmup 5"
run "4 contexts (default)" $B
run "3 contexts" $B --contexts 3
run "5 contexts" $B --contexts 5
run "6 contexts" $B --contexts 6
run "4 contexts, no split" env MI355X_BZ2_NO_SPLIT=1 $B
run "5 contexts, no split, 16 queues" env GPU_MAX_HW_QUEUES=16 MI355X_BZ2_NO_SPLIT=1 $B --contexts 5
run "4 contexts, walks side by side" env MI355X_BZ2_WALK_SERIAL=0 $B
run "4 contexts, walk wgs 64" env MI355X_BZ2_WALK_WGS=64 $B
