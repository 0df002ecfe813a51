#!/bin/bash
# Developer probe: a rank's share of the file at N = 8 (310 blocks) and N = 4 (630) on one GPU, by contexts and hardware queues
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['config']['blocks_rank0'], 'blocks')"; }
S4="python bench.py --no-cpu-baseline --no-host-output --total-bytes 536870912 --base-bytes 53687091 --steps 24 --warmup 8"
S8="python bench.py --no-cpu-baseline --no-host-output --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 10"
run "N=8 share, 4 contexts" $S8
for q in 8 16; do for c in 4 5 6 8; do
run "N=8 share, $c contexts, $q queues" env GPU_MAX_HW_QUEUES=$q $S8 --contexts $c
done; done
run "N=4 share, 4 contexts" $S4
run "N=4 share, 5 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $S4 --contexts 5
run "N=4 share, 6 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $S4 --contexts 6
