#!/bin/bash
# Issue / LDS counters per kernel of one NO_SPLIT step (every kernel launched once over all 2 560 blocks):
# which pipe bounds a kernel that is neither HBM- nor MFMA-bound.  Separate --pmc passes, no tracing beside them.
# Usage: tools/sq_counters.sh <tag>   -> gpurun_out/<tag>_sq_counters.json
set -o pipefail
TAG=${1:-rXX}
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
export MI355X_BZ2_NO_SPLIT=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/sq_${TAG}_$i
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq_${TAG}_$i -- python3 $R/bench.py --steps 1 --warmup 1 --contexts 1 --resident --no-cpu-baseline > $R/gpurun_out/sq_${TAG}_$i.log 2>&1
    rc=$?; echo "pass $i ($set) exit=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(f"gpurun_out/sq_{tag}_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("bz2gpu::", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[(k, row["Counter_Name"])] += 1
out = {}
for k, c in tot.items():
    if not k.startswith("k_"): continue
    out[k] = {name: v / calls[(k, name)] for name, v in c.items()}
    out[k]["launches"] = max(calls[(k, n)] for n in c)
json.dump(out, open(f"gpurun_out/{tag}_sq_counters.json", "w"), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get("SQ_BUSY_CYCLES", 0)):
    print(k, {n: f"{v:.3g}" for n, v in sorted(out[k].items())})
PY
