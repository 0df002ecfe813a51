#!/bin/bash
# kernel trace of the rank-share configuration (310 blocks per step) and of the default bench; timelines to gpurun_out/
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_share $R/gpurun_out/trace_full
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_share -- python3 $R/bench.py --total-bytes 268435456 --base-bytes 26843545 --steps 24 --warmup 4 --contexts ${CONTEXTS:-4} --no-cpu-baseline > $R/gpurun_out/trace_share.log 2>&1 || exit 1
f=$(find $R/gpurun_out/trace_share -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 60 200 -1 > $R/gpurun_out/timeline_share.txt
[ "$1" = share ] && { rm -rf $R/gpurun_out/trace_share; exit 0; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_full -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $R/gpurun_out/trace_full.log 2>&1 || exit 1
f=$(find $R/gpurun_out/trace_full -name "*kernel_trace.csv" | head -1)
python3 $R/tools/timeline.py $f 240 200 -1 > $R/gpurun_out/timeline_full.txt
rm -rf $R/gpurun_out/trace_share $R/gpurun_out/trace_full
