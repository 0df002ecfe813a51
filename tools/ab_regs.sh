#!/bin/bash
# Developer probe: register budgets of k_hscan<1> / k_hsym / k_mtf / k_link2 (MI355X_BZ2_REGS), variants side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_hscan', k.get('k_hscan'), 'k_mtf', k.get('k_mtf<272>'), k.get('k_mtf<144>'), 'k_link2', k.get('k_link2'), 'k_hsym', k.get('k_hsym'), 'k_walk', k.get('k_walk'), 'k_emit', k.get('k_emit'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 4"
for v in "scan=4,sym=128,mtf=4,link=4" "scan=4,sym=256,mtf=4,link=4" "scan=4,sym=512,mtf=4,link=4"; do
run "regs $v" env MI355X_BZ2_REGS=$v $B
done
run "segments short" env MI355X_BZ2_SEGMENTS=short $B
run "segments long" env MI355X_BZ2_SEGMENTS=long $B
run "segments long, sym=256" env MI355X_BZ2_SEGMENTS=long MI355X_BZ2_REGS=sym=256 $B
run "5 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $B --contexts 5
run "6 contexts, 16 queues" env GPU_MAX_HW_QUEUES=16 $B --contexts 6
