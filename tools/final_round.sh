#!/bin/bash
# The measurements quoted in DESIGN.md / README.md, one after the other on ONE box; everything into gpurun_out/ (copy what is
# to be judged into profiles/).  Usage: tools/final_round.sh <tag> [bench] [probes] [reader] [configs] [profile]
set -o pipefail
TAG=${1:-rXX}; shift
WHAT=${@:-bench probes reader configs profile}
mkdir -p gpurun_out
for w in $WHAT; do
  case $w in
    bench)
      python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
      echo "bench: $(cut -c1-200 gpurun_out/${TAG}_bench.json)" ;;
    probes)
      python tools/huff_probe.py scan > gpurun_out/${TAG}_latency.txt 2>&1 || exit 1
      bash tools/share_probe.sh > gpurun_out/${TAG}_share.txt 2>&1
      { python bench.py --no-cpu-baseline --no-host-output --total-bytes 1073741824 --base-bytes 107374182 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('N=2 share (1280 blocks per step), 4 contexts:', d['ms_per_step'], 'ms/step')"; } >> gpurun_out/${TAG}_share.txt
      cat gpurun_out/${TAG}_share.txt ;;
    reader)
      python tools/reader_probe2.py 512,512 4 > gpurun_out/${TAG}_reader.txt 2>&1 || exit 1
      PROBE_WARMUP=1 python tools/reader_probe2.py 512,512 4 >> gpurun_out/${TAG}_reader.txt 2>&1 || exit 1
      python tools/reader_probe2.py 1024,1024 4 >> gpurun_out/${TAG}_reader.txt 2>&1 || exit 1
      MI355X_BZ2_INPUT_BUDGET=1000000 python tools/reader_probe2.py 512 4 >> gpurun_out/${TAG}_reader.txt 2>&1 || exit 1
      tail -12 gpurun_out/${TAG}_reader.txt ;;
    configs)
      python tools/bench_configs.py > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err || exit 1
      tail -3 gpurun_out/${TAG}_configs.err ;;
    profile)
      bash tools/profile_round.sh $TAG default nosplit traffic config3 > gpurun_out/${TAG}_profile.log 2>&1 || exit 1
      tail -30 gpurun_out/${TAG}_profile.log ;;
  esac
done
echo done
