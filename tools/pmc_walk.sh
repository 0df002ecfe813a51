#!/bin/bash
# Memory-side counters of k_walk (every kernel launched once over the bench's batch) and of the gather micro-benchmark, to
# see what the walk's gathers wait for.  Separate --pmc passes, nothing else traced.
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
export MI355X_BZ2_NO_SPLIT=1
i=0
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
            "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcw_$i $R/gpurun_out/pmcu_$i
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcw_$i -- python3 $R/bench.py --steps 1 --warmup 1 --contexts 1 --resident --no-cpu-baseline --no-host-output > $R/gpurun_out/pmcw_$i.log 2>&1
  echo "walk pass $i exit=$?"
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcu_$i -- $R/tools/ubench/bin/gather_rate > $R/gpurun_out/pmcu_$i.log 2>&1
  echo "ubench pass $i exit=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for what, pat, match in (("k_walk", "gpurun_out/pmcw_*/**/*counter_collection.csv", "k_walk<"), ("ubench k_chase<2>", "gpurun_out/pmcu_*/**/*counter_collection.csv", "k_chase<2>")):
    agg = collections.defaultdict(float); calls = collections.Counter()
    for f in sorted(glob.glob(pat, recursive=True)):
        for row in csv.DictReader(open(f)):
            if match in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
    print(what)
    for c in sorted(agg): print(f"  {c:36s} {agg[c] / calls[c]:18.0f} per launch ({calls[c]} launches)")
PY
