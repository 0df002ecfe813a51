#!/bin/bash
# HBM traffic of one bench step via PMC (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_$c.log 2>&1
  echo "pass $c exit=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob("gpurun_out/traffic_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
out = {}
for k, d in tot.items():
    if "bz2gpu" in k:
        out[k] = {c: v / max(1, calls[(k, c)]) for c, v in d.items()}   # per launch, raw counter units (KiB per rocprof)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/traffic_raw.json", "w"), indent=1)
PY
