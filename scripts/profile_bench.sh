#!/bin/bash
# Reproduces the committed profiles/: rocprofv3 kernel trace + stats of bench.py, then the HBM
# counters in SEPARATE passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never combined with other trace domains).  Run on the GPU box from the repo root:
#   bash scripts/profile_bench.sh [tag]          (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 20 --warmup 3 --no-cpu"
run() { timeout 180 "$@" < /dev/null; }
run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/pmc_fetch.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/pmc_write.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/pmc_sq.log 2>&1
# the HBM-resident point (4M nodes)
run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_2048 -o bench -- python3 bench.py --rows 2048 --cols 2048 --steps 5 --warmup 2 --no-cpu > $OUT/bench_trace_2048.log 2>&1
run rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_2048 -o b -- python3 bench.py --rows 2048 --cols 2048 --steps 3 --warmup 1 --no-cpu > $OUT/pmc_fetch_2048.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_2048 -o b -- python3 bench.py --rows 2048 --cols 2048 --steps 3 --warmup 1 --no-cpu > $OUT/pmc_write_2048.log 2>&1
python3 scripts/summarize_profile.py $OUT $TAG
