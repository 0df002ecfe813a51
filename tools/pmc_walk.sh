#!/bin/bash
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcw_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcw_$tag.log 2>&1
  echo "pass $tag exit=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcw_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k,row["Counter_Name"])] += 1
    for k, d in agg.items():
        if "walk" in k or "bwt" in k:
            print(k[:36], {c: round(v / calls[(k,c)]) for c, v in d.items()})
PY
