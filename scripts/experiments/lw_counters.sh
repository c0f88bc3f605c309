#!/bin/bash
# Memory-pipeline counters of the likelihood-weighting kernels on config 5 (one rocprofv3 --pmc pass per group).
# GPU box, from the repo root:  bash scripts/experiments/lw_counters.sh [samples]
export TMPDIR=/tmp
OUT=gpurun_out/lw_counters
mkdir -p $OUT
N=${1:-2000000}
i=0
while read -r group; do
    [ -z "$group" ] && continue
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/g$i -o c -- python3 scripts/time_lw.py $N > $OUT/g$i.log 2>&1 < /dev/null || echo "group $i failed: $group"
done <<'GROUPS'
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY
TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum
SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
GROUPS
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in sorted(glob.glob("$OUT/g*/**/c_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:28]
        if "lw_" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    
for k in tot:
    print(k)
    for c, v in sorted(tot[k].items()): print(f"   {c:44s} {v:.4g}")
PY
