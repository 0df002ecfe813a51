#!/bin/bash
# Developer probe: FETCH_SIZE of k_walk per step of the bench as it runs (four contexts: the crowd's kernel choices), for
# forms of the walk side by side on ONE box.  One --pmc pass per form, nothing traced beside it.
# a form = "<name> <library or -> <walk wgs or -> <chunk or ->"
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1 lib=$2 wgs=$3 chunk=$4
    unset MI355X_BZ2_LIBRARY MI355X_BZ2_WALK_WGS MI355X_BZ2_WALK_CHUNK
    [ "$lib" != - ] && export MI355X_BZ2_LIBRARY=$R/indexed_bzip2_amd/$lib
    [ "$wgs" != - ] && export MI355X_BZ2_WALK_WGS=$wgs
    [ "$chunk" != - ] && export MI355X_BZ2_WALK_CHUNK=$chunk
    rm -rf $R/gpurun_out/pmcw_$name
    timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcw_$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-output > $R/gpurun_out/pmcw_$name.log 2>&1
    local rc=$?
    python3 - "$name" "$R" <<'PY'
import csv, glob, sys, collections
name, root = sys.argv[1], sys.argv[2]
tot = collections.Counter(); calls = collections.Counter()
for f in glob.glob(f"{root}/gpurun_out/pmcw_{name}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bz2gpu::", "")
        tot[k] += float(row["Counter_Value"]); calls[k] += 1
steps = max(1, calls["k_crc"])
print(name, "k_walk FETCH_SIZE per step: %.2f GB" % (sum(v for k, v in tot.items() if k.startswith("k_walk") and "plan" not in k) * 1024 / steps / 1e9),
      "k_emit x2: %.2f GB" % (sum(v for k, v in tot.items() if k.startswith("k_emit")) * 2048 / steps / 1e9), "steps", steps, flush=True)
PY
    rm -rf $R/gpurun_out/pmcw_$name
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name: timed out"; exit $rc; fi
}
while [ $# -ge 4 ]; do run $1 $2 $3 $4 || exit 1; shift 4; done
