#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --steps 16 --warmup 4"
for seg in 1 2 4; do run "segments per lane $seg" env MI355X_BZ2_WALK_SEGMENTS=$seg $B; done
for seg in 2 4; do run "segments per lane $seg, 32 wgs" env MI355X_BZ2_WALK_SEGMENTS=$seg MI355X_BZ2_WALK_WGS=32 $B; done
run "segments per lane 1" env MI355X_BZ2_WALK_SEGMENTS=1 $B
