#!/bin/bash
# Run GPU steps in order; a step that times out / is killed ends the call (no further GPU work behind a hung step),
# an ordinary failure (assertion, non-zero exit) is recorded and the next step still runs.
# Usage: tools/gpu_step.sh <tag> '<cmd1>' '<cmd2>' ...   (each cmd gets `timeout -k 10 $STEP_TIMEOUT`)
set -o pipefail
TAG=$1; shift
mkdir -p gpurun_out
i=0
for cmd in "$@"; do
    i=$((i+1))
    log=gpurun_out/${TAG}_step$i.log
    echo "== step $i: $cmd" | tee $log
    timeout -k 10 ${STEP_TIMEOUT:-700} bash -c "$cmd" >> $log 2>&1
    rc=$?
    echo "== step $i exit=$rc" | tee -a $log
    tail -n ${TAIL:-15} $log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $i timed out: stopping"; exit $rc; fi
done
exit 0
