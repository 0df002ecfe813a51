#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --steps 16 --warmup 4"
run "full" $B
run "full" $B
run "full 96 wgs" env MI355X_BZ2_WALK_WGS=96 $B
run "full 48 wgs" env MI355X_BZ2_WALK_WGS=48 $B
run "share" python bench.py --no-cpu-baseline --total-bytes 268435456 --base-bytes 26843545 --steps 60 --warmup 8
