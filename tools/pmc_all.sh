#!/bin/bash
# Instruction and wave-cycle totals of every pipeline kernel for one bench step (counters serialize the kernels).
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcall
timeout -k 10 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmcall -- python3 $R/bench.py --steps 1 --warmup 1 --contexts 1 --no-cpu-baseline > $R/gpurun_out/pmcall.log 2>&1
echo "exit=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmcall/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bz2gpu::", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
steps = max(1, calls[("k_crc", "SQ_WAVES")])
print(f"decode_batch calls profiled: {steps}")
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
print(f"{'kernel':18s}" + "".join(f"{n[3:]:>16s}" for n in names))
grand = collections.defaultdict(float)
for k in sorted(tot):
    if k.startswith("k_bswap") or k.startswith("k_find") or k.startswith("__amd"): continue
    print(f"{k:18s}" + "".join(f"{tot[k][n] / steps / 1e6:16.1f}" for n in names))
    for n in names: grand[n] += tot[k][n] / steps
print(f"{'TOTAL (millions)':18s}" + "".join(f"{grand[n] / 1e6:16.1f}" for n in names))
PY
