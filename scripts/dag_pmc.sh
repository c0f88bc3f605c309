#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/dagpmc
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o c -- python3 bench.py --no-extras --no-cpu --workload dag --steps 5 --warmup 2 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -o c -- python3 bench.py --no-extras --no-cpu --workload dag --steps 5 --warmup 2 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv,glob,statistics,collections
for d in ("a","b"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/dagpmc/{d}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "bp_sweep_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print(d,k,len(v),statistics.median(v))
PY
tail -2 $OUT/a.log
