#!/bin/bash
# Developer probe: k_hscan<1> grid caps (LDS left free for k_hsym / k_mtf of the batches beside it), on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms_in_the_crowd']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'in the crowd: k_hsym', k.get('k_hsym'), 'k_hscan', k.get('k_hscan'), 'k_mtf', k.get('k_mtf<272>'), k.get('k_mtf<144>'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 4"
run "default" $B
for g in 2048 1792 1536 1280 1024; do run "scan grid $g" env MI355X_BZ2_SCAN_GRID=$g $B; done
for g in 1792 1280; do run "scan grid $g, sym=128" env MI355X_BZ2_SCAN_GRID=$g MI355X_BZ2_REGS=sym=128 $B; done
run "scan grid 1536, 5 contexts, 16 queues" env MI355X_BZ2_SCAN_GRID=1536 GPU_MAX_HW_QUEUES=16 $B --contexts 5
