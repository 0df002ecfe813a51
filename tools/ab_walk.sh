#!/bin/bash
# Developer probe: the walk's workgroups per XCD and segment form, side by side on ONE box
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s', 'alone: k_walk', k.get('k_walk'), 'k_emit', k.get('k_emit'), 'k_link2', k.get('k_link2'), 'k_hsym', k.get('k_hsym'), 'sum', d['roofline'].get('kernel_ms_sum_alone'))"; }
B="python bench.py --no-cpu-baseline --no-host-output --steps 16 --warmup 4"
run "default" $B
for w in 32 64 128 256; do run "walk wgs $w" env MI355X_BZ2_WALK_WGS=$w $B; done
run "walks side by side" env MI355X_BZ2_WALK_SERIAL=0 $B
run "walks side by side, 32 wgs" env MI355X_BZ2_WALK_SERIAL=0 MI355X_BZ2_WALK_WGS=32 $B
run "segments long" env MI355X_BZ2_SEGMENTS=long $B
run "chunk 1024" env MI355X_BZ2_WALK_CHUNK=1024 $B
run "one context resident" $B --contexts 1 --resident
