#!/bin/bash
# SQ counters of the sweep kernel on one small network: bash scripts/net_pmc.sh alarm_shaped
export TMPDIR=/tmp
NET=${1:-alarm_shaped}
OUT=gpurun_out/netpmc_$NET
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o c -- python3 scripts/run_net.py $NET > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -o c -- python3 scripts/run_net.py $NET > $OUT/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 scripts/run_net.py $NET > $OUT/t.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,statistics,collections,sys
out=sys.argv[1]
for d in ("a","b"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "bp_sweep" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(d,k,len(v),statistics.median(v))
for f in glob.glob(f"{out}/t/**/*kernel_stats.csv",recursive=True):
    print(open(f).read()[:1500])
PY
tail -1 $OUT/t.log
