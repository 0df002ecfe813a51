#!/bin/bash
# Developer probe: variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
B="python bench.py --no-cpu-baseline --steps 12 --warmup 3"
run "full 4ctx" $B
run "turns mtf" env MI355X_BZ2_TURNS=0x802 $B
run "turns bwt" env MI355X_BZ2_TURNS=0x4 $B
run "turns link2" env MI355X_BZ2_TURNS=0x10 $B
run "turns emit" env MI355X_BZ2_TURNS=0x20 $B
run "turns hsym" env MI355X_BZ2_TURNS=0x2000 $B
run "turns hscan" env MI355X_BZ2_TURNS=0x1000 $B
run "turns mtf+bwt+link2+emit" env MI355X_BZ2_TURNS=0x836 $B
run "turns all but hscan" env MI355X_BZ2_TURNS=0x2836 $B
run "full 4ctx" $B
