#!/bin/bash
# Developer probe: rounds of doubling in k_hscan<1> (MI355X_BZ2_SCAN_TUNE: 4 = three rounds for spans of one or two groups, 8 = three always, 16 = four always)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_hscan', k.get('k_hscan'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
for w in silesia urandom; do
B="python bench.py --workload $w --no-cpu-baseline --no-host-output --steps 10 --warmup 5"
for t in 0 4 8 16; do run "$w tune $t" env MI355X_BZ2_SCAN_TUNE=$t $B; done
done
