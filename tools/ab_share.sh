#!/bin/bash
# Developer probe: a rank's share of the file at N = 2 and N = 8 on one GPU, by the scan's form
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['config']['blocks_rank0'], 'blocks')"; }
S2="python bench.py --no-cpu-baseline --no-host-output --total-bytes 1073741824 --base-bytes 107374182 --steps 16 --warmup 4"
S8="python bench.py --no-cpu-baseline --no-host-output --total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8"
for w in 0 1 4 8; do run "N=2 share, scan waves $w" env MI355X_BZ2_SCAN_WAVES=$w $S2; done
for w in 0 1 4 8; do run "N=8 share, scan waves $w" env MI355X_BZ2_SCAN_WAVES=$w $S8; done
run "N=8 share, bwt split 4" env MI355X_BZ2_BWT_SPLIT=4 $S8
run "N=8 share, bwt split 1" env MI355X_BZ2_BWT_SPLIT=1 $S8
run "N=8 share, walk wgs 64 chunk 256" env MI355X_BZ2_WALK_WGS=64 MI355X_BZ2_WALK_CHUNK=256 $S8
run "N=8 share, 3 contexts" $S8 --contexts 3
run "N=2 share, walk wgs 64 chunk 256" env MI355X_BZ2_WALK_WGS=64 MI355X_BZ2_WALK_CHUNK=256 $S2
