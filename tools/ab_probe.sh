#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --steps 12 --warmup 3"
S="--total-bytes 268435456 --base-bytes 26843545 --steps 60 --warmup 8"
run "full 4ctx (default: walks in turn, 64)" $B
run "full 4ctx walks not in turn, 256" env MI355X_BZ2_WALK_SERIAL=0 MI355X_BZ2_WALK_WGS=256 $B
run "full 3ctx" $B --contexts 3
run "full 2ctx" $B --contexts 2
run "share 4ctx" python bench.py --no-cpu-baseline $S
run "share 4ctx walks not in turn, 256" env MI355X_BZ2_WALK_SERIAL=0 MI355X_BZ2_WALK_WGS=256 python bench.py --no-cpu-baseline $S
run "half 4ctx" python bench.py --no-cpu-baseline --total-bytes 1073741824 --base-bytes 107374182 --steps 16 --warmup 4
