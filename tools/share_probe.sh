#!/bin/bash
# Developer probe: one rank's share of the bench file at N = 8 (310 blocks per step) on one GPU, by contexts and k_hscan waves per block
for c in ${CONTEXTS:-4 5}; do for w in ${WAVES:-0}; do
echo -n "contexts=$c scan_waves=$w: "; MI355X_BZ2_SCAN_WAVES=$w python bench.py --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8 --contexts $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', d['config']['blocks_rank0'], 'blocks')"
done; done
