#!/bin/bash
# HBM traffic of one bench step via PMC (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_$c.log 2>&1
  echo "pass $c exit=$?"
done
cd $R
python3 - <<'PY'
# Per-step HBM traffic: sum over ALL launches of the pipeline kernels / number of decode_batch calls (= k_crc launches).
# Units and corrections as in /opt/skills/guides/MI355X_MICROARCH.md: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB;
# on gfx950 FETCH_SIZE counts half of the bytes (calibrated here on k_bswap32, which reads and writes the input once).
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob("gpurun_out/traffic_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])] += 1
steps = max(1, calls[("bz2gpu::k_crc", "FETCH_SIZE")])
pipeline = [k for k in tot if "bz2gpu" in k and "k_bswap32" not in k and "k_find_magic" not in k]
fetch = sum(tot[k]["FETCH_SIZE"] for k in pipeline) * 1024 * 2 / steps
write = sum(tot[k]["WRITE_SIZE"] for k in pipeline) * 1024 / steps
bs = tot.get("bz2gpu::k_bswap32", {})
out = {"hbm_bytes_per_step": int(fetch + write), "fetch_bytes_x2": int(fetch), "write_bytes": int(write), "steps_profiled": steps,
       "calibration_k_bswap32": {"fetch_kib": bs.get("FETCH_SIZE", 0) / max(1, calls[("bz2gpu::k_bswap32", "FETCH_SIZE")]),
                                 "write_kib": bs.get("WRITE_SIZE", 0) / max(1, calls[("bz2gpu::k_bswap32", "WRITE_SIZE")])},
       "per_kernel_bytes_per_step": {k: {"fetch_x2": int(tot[k]["FETCH_SIZE"] * 2048 / steps), "write": int(tot[k]["WRITE_SIZE"] * 1024 / steps),
                                         "launches_per_step": calls[(k, "FETCH_SIZE")] / steps} for k in sorted(pipeline)}}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/traffic_latest.json", "w"), indent=1)
PY
