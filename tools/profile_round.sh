#!/bin/bash
# The rocprofv3 artefacts of a round, all under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_kernel_stats_default.csv    rocprofv3 --kernel-trace --stats of `bench.py` as the driver runs it
#   <tag>_kernel_stats_nosplit.csv    the same with MI355X_BZ2_NO_SPLIT=1 --contexts 1: every kernel launched ONCE per step
#                                     over all 2 560 blocks, nothing overlapped (the per-kernel table of DESIGN.md)
#   <tag>_traffic.json                FETCH_SIZE / WRITE_SIZE per kernel and step (separate --pmc passes), with the
#                                     correction of tools/fetch_calib.sh
#   <tag>_config3_*                   the same three for config 3 (2 GiB of random bytes)
# Usage: tools/profile_round.sh <tag> [default|nosplit|traffic|config3 ...]
set -o pipefail
TAG=${1:-rXX}; shift
WHAT=${@:-default nosplit traffic config3}
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
stats() {   # stats <name> <env...> -- <args of the program after python3>
    local name=$1; shift
    rm -rf $R/gpurun_out/prof_${TAG}_$name
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$name -- python3 "$@" > $R/gpurun_out/prof_${TAG}_$name.log 2>&1
    local rc=$?; echo "stats $name exit=$rc"
    local f=$(find $R/gpurun_out/prof_${TAG}_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_kernel_stats_$name.csv && head -16 "$f"
    return $rc
}
pmc() {     # pmc <name> <args...>: FETCH_SIZE and WRITE_SIZE passes
    local name=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $R/gpurun_out/pmc_${TAG}_${name}_$c
        timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name}_$c -- python3 "$@" > $R/gpurun_out/pmc_${TAG}_${name}_$c.log 2>&1
        echo "pmc $name $c exit=$?"
    done
}
for w in $WHAT; do
  case $w in
    default) stats default $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-output || exit 1 ;;
    nosplit) MI355X_BZ2_NO_SPLIT=1 stats nosplit $R/bench.py --steps 3 --warmup 1 --contexts 1 --resident --no-cpu-baseline --no-host-output || exit 1 ;;
    traffic) pmc bench $R/bench.py --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --no-host-output ;;
    config3) MI355X_BZ2_NO_SPLIT=1 stats config3 $R/bench.py --workload urandom --steps 3 --warmup 1 --contexts 1 --resident --no-cpu-baseline --no-host-output || exit 1
             pmc config3 $R/bench.py --workload urandom --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --no-host-output ;;
  esac
done
cd $R
python3 - "$TAG" <<'PY'
# Per-step traffic: sum over ALL launches of the pipeline kernels / number of decode_batch calls (= k_crc launches).
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  Corrections (tools/fetch_calib.sh on this machine, committed as
# profiles/r02_fetch_calib.json): FETCH_SIZE counts HALF of the bytes of coalesced reads (4 or 16 B per lane) and EXACTLY
# 64 B per random 4-byte gather that misses; WRITE_SIZE is exact.  So k_walk (gathers) takes factor 1, everything else 2.
import csv, glob, collections, json, sys
tag = sys.argv[1]
for name in ("bench", "config3"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    files = sorted(glob.glob(f"gpurun_out/pmc_{tag}_{name}_*/**/*counter_collection.csv", recursive=True))
    if not files:
        continue
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    steps = max(1, calls[("bz2gpu::k_crc", "FETCH_SIZE")])
    pipeline = [k for k in tot if "bz2gpu" in k and "k_find_magic" not in k]
    factor = lambda k: 1 if ("k_walk" in k and "plan" not in k) else 2
    fetch = sum(tot[k]["FETCH_SIZE"] * factor(k) for k in pipeline) * 1024 / steps
    fetch_hi = sum(tot[k]["FETCH_SIZE"] for k in pipeline) * 2048 / steps
    write = sum(tot[k]["WRITE_SIZE"] for k in pipeline) * 1024 / steps
    out = {"what": name, "hbm_bytes_per_step": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
           "fetch_bytes_if_everything_x2": int(fetch_hi), "steps_profiled": steps,
           "correction": "FETCH_SIZE x 2 for every kernel except k_walk (x 1: its reads are 4-byte gathers, counted exactly, "
                         "see profiles/r02_fetch_calib.json); WRITE_SIZE x 1",
           "per_kernel_bytes_per_step": {k: {"fetch": int(tot[k]["FETCH_SIZE"] * factor(k) * 1024 / steps), "write": int(tot[k]["WRITE_SIZE"] * 1024 / steps),
                                             "launches_per_step": calls[(k, "FETCH_SIZE")] / steps} for k in sorted(pipeline)}}
    json.dump(out, open(f"gpurun_out/{tag}_traffic_{name}.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel_bytes_per_step"}))
    for k, v in out["per_kernel_bytes_per_step"].items():
        print(f"  {k:28s} fetch {v['fetch'] / 1e9:8.2f} GB  write {v['write'] / 1e9:8.2f} GB  launches {v['launches_per_step']}")
PY
