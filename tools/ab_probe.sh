#!/bin/bash
# Developer probe: bench variants side by side on ONE box (box-to-box variance is +-5 %)
run() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'MB/s')"; }
S="--total-bytes 268435456 --base-bytes 26843545 --steps 40 --warmup 8 --no-cpu-baseline"
for rep in 1 2; do
run "share 4ctx prefetch" python bench.py $S --contexts 4
run "share 4ctx no prefetch" env BENCH_NO_PREFETCH=1 python bench.py $S --contexts 4
run "share 4ctx resident" python bench.py $S --contexts 4 --resident
done
run "share 5ctx prefetch" python bench.py $S --contexts 5
run "full 3ctx prefetch" python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run "full 3ctx no prefetch" env BENCH_NO_PREFETCH=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run "full 4ctx prefetch" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --contexts 4
run "full 5ctx prefetch" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --contexts 5
