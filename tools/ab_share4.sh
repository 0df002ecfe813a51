#!/bin/bash
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['config']['blocks_rank0'], 'blocks')"; }
S4="python bench.py --no-cpu-baseline --no-host-output --total-bytes 536870912 --base-bytes 53687091 --steps 24 --warmup 6"
S3="python bench.py --no-cpu-baseline --no-host-output --total-bytes 805306368 --base-bytes 80530636 --steps 20 --warmup 6"
for w in 1 4; do run "N=4 share (640 blocks), scan waves $w" env MI355X_BZ2_SCAN_WAVES=$w $S4; done
for w in 1 4; do run "960 blocks, scan waves $w" env MI355X_BZ2_SCAN_WAVES=$w $S3; done
