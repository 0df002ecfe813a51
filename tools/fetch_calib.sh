#!/bin/bash
# What FETCH_SIZE / WRITE_SIZE report for known byte counts per access shape (tools/ubench/fetch_calib.hip): the
# correction factors applied to the pipeline kernels' traffic come from here.  Output: gpurun_out/fetch_calib.json
R=$PWD; mkdir -p gpurun_out tools/ubench/bin
[ -x tools/ubench/bin/fetch_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/ubench/bin/fetch_calib tools/ubench/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf $R/gpurun_out/calib_$tag
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/calib_$tag -- $R/tools/ubench/bin/fetch_calib > $R/gpurun_out/calib_$tag.log 2>&1
  echo "pass $c exit=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
expect = None
for f in glob.glob("gpurun_out/calib_*.log"):
    for line in open(f):
        if line.startswith("{"):
            expect = json.loads(line)
tot = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/calib_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("calib_"):
            tot[k][row["Counter_Name"]] = tot[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
out = {"expected": expect, "counters": tot, "derived": {}}
for k, c in tot.items():
    d = {}
    if "FETCH_SIZE" in c: d["FETCH_SIZE_bytes(KiB*1024)"] = c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c: d["WRITE_SIZE_bytes(KiB*1024)"] = c["WRITE_SIZE"] * 1024
    if "TCC_EA0_RDREQ_sum" in c:
        d["rdreq"] = c["TCC_EA0_RDREQ_sum"]; d["rdreq_32B"] = c.get("TCC_EA0_RDREQ_32B_sum", 0)
    e = (expect or {}).get(k, {})
    if "read_bytes" in e and "FETCH_SIZE" in c: d["true_bytes / FETCH_SIZE"] = e["read_bytes"] / (c["FETCH_SIZE"] * 1024)
    if "gathers" in e and "FETCH_SIZE" in c: d["FETCH_SIZE bytes per gather"] = c["FETCH_SIZE"] * 1024 / e["gathers"]
    if "gathers" in e and "TCC_EA0_RDREQ_sum" in c: d["read requests per gather"] = c["TCC_EA0_RDREQ_sum"] / e["gathers"]
    if "write_bytes" in e and "WRITE_SIZE" in c: d["true_bytes / WRITE_SIZE"] = e["write_bytes"] / (c["WRITE_SIZE"] * 1024)
    out["derived"][k] = d
print(json.dumps(out["derived"], indent=1))
json.dump(out, open("gpurun_out/fetch_calib.json", "w"), indent=1)
PY
