#!/bin/bash
# Developer probe: the scan's form for big batches, both workloads, side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_hscan', k.get('k_hscan'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
for w in urandom silesia; do
B="python bench.py --no-cpu-baseline --no-host-output --steps 12 --warmup 4 --workload $w"
run "$w default" $B
run "$w spec<4> for all" env MI355X_BZ2_SCAN_WAVES=4 $B
run "$w spec<8> for all" env MI355X_BZ2_SCAN_WAVES=8 $B
run "$w scan regs 2" env MI355X_BZ2_REGS=scan=2 $B
run "$w scan regs 5" env MI355X_BZ2_REGS=scan=5 $B
done
