#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --steps 16 --warmup 4"
run "default (1 walk at a time, 64)" $B
for w in 32 48 64; do run "2 walks at a time, $w wgs" env MI355X_BZ2_WALK_LANES=2 MI355X_BZ2_WALK_WGS=$w $B; done
run "3 walks at a time, 32 wgs" env MI355X_BZ2_WALK_LANES=3 MI355X_BZ2_WALK_WGS=32 $B
run "1 walk, 96" env MI355X_BZ2_WALK_WGS=96 $B
run "default" $B
