#!/bin/bash
# Likelihood weighting on config 5: samples per second and the two kernels' durations from rocprofv3, under environment switches.
# GPU box, from the repo root:  bash scripts/experiments/lw_knobs.sh "1 0" [samples]     (values of BN_LW_SMALL: 0 = the generic kernel)
export TMPDIR=/tmp
OUT=gpurun_out/lw_knobs
mkdir -p $OUT
for k in ${1:-1 0}; do
    export BN_LW_SMALL=$k
    python3 scripts/time_lw.py ${2:-2000000} 2>&1 | grep samples | sed "s/^/BN_LW_SMALL=$k  /"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$k -o t -- python3 scripts/time_lw.py ${2:-2000000} > $OUT/k$k.log 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/k$k/**/t_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lw_" in r["Name"]: print("BN_LW_SMALL=$k ", r["Name"][:60], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
done
