#!/bin/bash
# The measurements quoted in DESIGN.md / README.md, one after the other on ONE box; everything into gpurun_out/
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || exit 1
echo "bench: $(cut -c1-160 gpurun_out/r02_bench.json)"
python bench.py --contexts 3 --no-cpu-baseline > gpurun_out/r02_bench_3contexts.json 2>/dev/null || exit 1
python tools/huff_probe.py scan > gpurun_out/r02_latency.txt 2>&1 || exit 1
bash tools/share_probe.sh > gpurun_out/r02_share.txt 2>&1
{ python bench.py --no-cpu-baseline --total-bytes 1073741824 --base-bytes 107374182 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('N=2 share (1280 blocks per step), 4 contexts:', d['ms_per_step'], 'ms/step')"; } >> gpurun_out/r02_share.txt
python tools/reader_probe2.py 512,512 4 > gpurun_out/r02_reader.txt 2>&1 || exit 1
PROBE_WARMUP=1 python tools/reader_probe2.py 512 4 >> gpurun_out/r02_reader.txt 2>&1
python tools/reader_probe2.py 1024,1024 4 >> gpurun_out/r02_reader.txt 2>&1
python tools/bench_configs.py 3 5 > gpurun_out/r02_configs.json 2> gpurun_out/r02_configs.err
echo done
