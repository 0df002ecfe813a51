#!/bin/bash
# Developer probe: instruction-cache requests / hits / misses per kernel of one NO_SPLIT step (every kernel launched once over
# all 2 560 blocks).  One --pmc pass, nothing traced beside it.
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
export MI355X_BZ2_NO_SPLIT=1
rm -rf $R/gpurun_out/pmc_icache
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $R/gpurun_out/pmc_icache -- python3 $R/bench.py --steps 1 --warmup 1 --contexts 1 --resident --no-cpu-baseline --no-host-output > $R/gpurun_out/pmc_icache.log 2>&1
rc=$?; echo "exit=$rc"; [ $rc -eq 124 ] && exit 124
python3 - "$R" <<'PY'
import csv, glob, collections, sys
root = sys.argv[1]
tot = collections.defaultdict(collections.Counter)
for f in glob.glob(f"{root}/gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bz2gpu::", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["SQC_ICACHE_REQ"]):
    req = v["SQC_ICACHE_REQ"] or 1
    print(f"{k:28s} req {req / 1e6:10.1f} M  hits {v['SQC_ICACHE_HITS'] / 1e6:10.1f} M  misses {v['SQC_ICACHE_MISSES'] / 1e6:9.1f} M  miss ratio {v['SQC_ICACHE_MISSES'] / req:6.3f}")
PY
rm -rf $R/gpurun_out/pmc_icache
