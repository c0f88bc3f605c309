#!/bin/bash
# Likelihood weighting on config 5 with the sampler's diagnostic switches (BN_LW_KNOBS), kernel durations from rocprofv3.
# GPU box, from the repo root:  bash scripts/experiments/lw_knobs.sh "0 1" [samples]
export TMPDIR=/tmp
OUT=gpurun_out/lw_knobs
mkdir -p $OUT
for k in ${1:-0}; do
    export BN_LW_KNOBS=$k
    python3 scripts/time_lw.py ${2:-2000000} 2>&1 | grep samples | sed "s/^/knobs=$k  /"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$k -o t -- python3 scripts/time_lw.py ${2:-2000000} > $OUT/k$k.log 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/k$k/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lw_" in r["Name"]: print("knobs=$k ", r["Name"][:60], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
done
