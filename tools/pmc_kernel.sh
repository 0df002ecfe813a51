#!/bin/bash
# SQ counters of ONE kernel of the pipeline.  Usage: tools/pmc_kernel.sh <kernel substring> [text|random] [blocks]
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
KERNEL=${1:-k_mtf}; KIND=${2:-text}; N=${3:-1024}
export MI355X_BZ2_NO_SPLIT=1
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
            "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmck_$i -- python3 $R/tools/scale_probe.py $KIND $N > $R/gpurun_out/pmck_$i.log 2>&1
  echo "pass $i exit=$?"
done
cd $R
KERNEL=$KERNEL python3 - <<'PY'
import csv, glob, collections, os
tot = collections.defaultdict(float); calls = collections.defaultdict(int)
for f in sorted(glob.glob("gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); calls[row["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]/calls[k]:16.0f}  (per launch, {calls[k]} launches)")
g = lambda k: tot[k]/max(1,calls[k])
w = g("SQ_WAVE_CYCLES")
print("wave-cycles per VALU inst:", w/max(1,g("SQ_INSTS_VALU")), " per LDS inst:", w/max(1,g("SQ_INSTS_LDS")), " per SALU:", w/max(1,g("SQ_INSTS_SALU")))
print("LDS latency (level/insts):", g("SQ_INST_LEVEL_LDS")/max(1,g("SQ_INSTS_LDS")))
print("waiting fraction (WAIT_INST_ANY/WAVE_CYCLES):", g("SQ_WAIT_INST_ANY")/max(1,w), " LDS wait:", g("SQ_WAIT_INST_LDS")/max(1,w))
PY
