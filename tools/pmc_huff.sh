#!/bin/bash
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
KIND=${1:-text}; N=${2:-1024}
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmch_$i -- python3 $R/tools/scale_probe.py $KIND $N > $R/gpurun_out/pmch_$i.log 2>&1
  echo "pass $i exit=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in sorted(glob.glob("gpurun_out/pmch_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "k_huff" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot): print(k, int(tot[k]))
w = tot["SQ_WAVE_CYCLES"]
print("LDS latency (level/insts):", tot["SQ_INST_LEVEL_LDS"]/max(1,tot["SQ_INSTS_LDS"]))
print("VMEM latency:", tot["SQ_INST_LEVEL_VMEM"]/max(1,tot["SQ_INSTS_VMEM_RD"]+tot["SQ_INSTS_VMEM_WR"]))
print("SMEM latency:", tot["SQ_INST_LEVEL_SMEM"]/max(1,tot["SQ_INSTS_SMEM"]))
print("IFETCH latency:", tot["SQ_IFETCH_LEVEL"]/max(1,tot["SQ_IFETCH"]), "ifetch per branch", tot["SQ_IFETCH"]/max(1,tot["SQ_INSTS_BRANCH"]))
print("wave_cycles per LDS inst (=per window-ish):", w/max(1,tot["SQ_INSTS_LDS"]))
PY
