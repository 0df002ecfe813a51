#!/bin/bash
# rocprofv3 PMC passes over scale_probe (counters only; never combined with traces)
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
KIND=${1:-text}; N=${2:-1024}
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_${KIND}_$tag -- python3 $R/tools/scale_probe.py $KIND $N > $R/gpurun_out/pmc_${KIND}_$tag.log 2>&1
  echo "pass $tag exit=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_${KIND}_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in agg.items():
        if "k_huff" in k or "k_mtf" in k or "k_walk" in k:
            print(k[:40], {c: int(v) for c, v in d.items()})
PY
