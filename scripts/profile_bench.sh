#!/bin/bash
# Reproduces the committed profiles/: rocprofv3 kernel trace + stats of bench.py, then the HBM
# counters in SEPARATE passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass;
# never combined with other trace domains).  Run on the GPU box from the repo root:
#   bash scripts/profile_bench.sh [tag]          (outputs under gpurun_out/prof_<tag>/)
# Every pass puts the program itself after `--` (no env / shell hop); switches travel as exported
# environment variables.  A pass that fails is recorded in failed_passes.txt and reported by the summary.
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
export BN_BENCH_NO_SETTLE=1   # bench.py's quarter second of untimed runs before the warm-up steps would be thousands of rows in every trace
mkdir -p $OUT
: > $OUT/failed_passes.txt
sha256sum bayesiannetwork_amd/libbn_mi355x.so | cut -d' ' -f1 > $OUT/lib_sha256.txt

pass() {  # name, then the rocprofv3 arguments
    local name=$1; shift
    timeout 300 rocprofv3 "$@" > $OUT/$name.log 2>&1 < /dev/null
    local rc=$?
    if [ $rc -ne 0 ]; then echo "$name rc=$rc" >> $OUT/failed_passes.txt; fi
}
trace() { pass $1 --kernel-trace --stats --output-format csv -d $OUT/$1 -o t -- python3 bench.py "${@:2}"; }
pmc() { pass $1 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o c -- python3 bench.py "${@:3}"; }

B="--no-extras --no-cpu"
# headline: 316x316 grid, default path (resident tiles, one launch per run)
export BN_MULTISWEEP=1
trace trace_grid316 $B --steps 20 --warmup 3
pmc fetch_grid316 FETCH_SIZE $B --steps 5 --warmup 2
pmc write_grid316 WRITE_SIZE $B --steps 5 --warmup 2
pmc sq_grid316 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --steps 5 --warmup 2
pmc sq2_grid316 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --steps 5 --warmup 2
# the same grid with one launch per sweep
export BN_MULTISWEEP=0
trace trace_grid316_launch $B --steps 20 --warmup 3
pmc fetch_grid316_launch FETCH_SIZE $B --steps 5 --warmup 2
pmc write_grid316_launch WRITE_SIZE $B --steps 5 --warmup 2
pmc sq_grid316_launch "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --steps 5 --warmup 2
export BN_MULTISWEEP=1
# config 2: 10 k-node DAG, default path (register-resident child tiles + parent items, one launch per run: bn_dag.hip)
trace trace_dag10k $B --workload dag --steps 20 --warmup 3
pmc fetch_dag10k FETCH_SIZE $B --workload dag --steps 5 --warmup 2
pmc write_dag10k WRITE_SIZE $B --workload dag --steps 5 --warmup 2
pmc sq_dag10k "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --workload dag --steps 5 --warmup 2
pmc sq2_dag10k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --workload dag --steps 5 --warmup 2
# ... and with one launch per sweep (the tile kernels)
export BN_DAG=0
trace trace_dag10k_launch $B --workload dag --steps 20 --warmup 3
pmc fetch_dag10k_launch FETCH_SIZE $B --workload dag --steps 5 --warmup 2
pmc write_dag10k_launch WRITE_SIZE $B --workload dag --steps 5 --warmup 2
unset BN_DAG
# config 0's network on the GPU: the ALARM-shaped net, one workgroup per run (bn_small.hip)
trace trace_alarm --no-cpu --workload alarm
pmc sq_alarm "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" --no-cpu --workload alarm
# a mid-size network (300 nodes, 38 workgroups) and a 10 000-node one (213 workgroups): several workgroups per run (bn_mid.hip)
trace trace_mid --no-cpu --workload mid
pmc sq_mid "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" --no-cpu --workload mid
# config 5: likelihood weighting
trace trace_lw $B --workload lw --steps 3
pmc fetch_lw FETCH_SIZE $B --workload lw --steps 2
pmc write_lw WRITE_SIZE $B --workload lw --steps 2
pmc sq_lw "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --workload lw --steps 2
pmc sq2_lw "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --workload lw --steps 2
# ... and the generic sampling kernel (mixed10k: arities 2-5), 10^6 samples per call
trace trace_lwgen $B --workload lwgen --steps 3
pmc fetch_lwgen FETCH_SIZE $B --workload lwgen --steps 2
pmc write_lwgen WRITE_SIZE $B --workload lwgen --steps 2
pmc sq_lwgen "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --workload lwgen --steps 2
pmc sq2_lwgen "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --workload lwgen --steps 2
# several evidence sets per call on the headline grid and on the 10 k-node DAG (bn_bp_run_batch_device, B = 16): bench.py --workload batch / dagbatch
trace trace_batch_grid316 $B --workload batch --steps 10 --warmup 3
pmc fetch_batch_grid316 FETCH_SIZE $B --workload batch --steps 4 --warmup 2
pmc write_batch_grid316 WRITE_SIZE $B --workload batch --steps 4 --warmup 2
pmc sq_batch_grid316 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --workload batch --steps 4 --warmup 2
pmc sq2_batch_grid316 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --workload batch --steps 4 --warmup 2
trace trace_batch_dag10k $B --workload dagbatch --steps 10 --warmup 3
pmc fetch_batch_dag10k FETCH_SIZE $B --workload dagbatch --steps 4 --warmup 2
pmc write_batch_dag10k WRITE_SIZE $B --workload dagbatch --steps 4 --warmup 2
pmc sq_batch_dag10k "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" $B --workload dagbatch --steps 4 --warmup 2
pmc sq2_batch_dag10k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $B --workload dagbatch --steps 4 --warmup 2
# the HBM-resident point (4M nodes)
trace trace_grid2048 $B --rows 2048 --cols 2048 --steps 3 --warmup 1
pmc fetch_grid2048 FETCH_SIZE $B --rows 2048 --cols 2048 --steps 2 --warmup 1
pmc write_grid2048 WRITE_SIZE $B --rows 2048 --cols 2048 --steps 2 --warmup 1
# what the vector ALUs issue on dependent integer code (the sampler's measured issue peak)
mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/experiments/valu_clock.hip -o build/valu_clock > /dev/null 2>&1 \
    && timeout 120 build/valu_clock > $OUT/valu_clock.txt 2>&1
python3 scripts/summarize_profile.py $OUT $TAG
