set -u
OUT=gpurun_out/prof_r03
export TMPDIR=/tmp
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mid -o t -- python3 bench.py --no-cpu --workload mid > $OUT/trace_mid.log 2>&1 < /dev/null || echo "trace_mid rc=$?" >> $OUT/failed_passes.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_mid -o c -- python3 bench.py --no-cpu --workload mid > $OUT/sq_mid.log 2>&1 < /dev/null || echo "sq_mid rc=$?" >> $OUT/failed_passes.txt
sha256sum bayesiannetwork_amd/libbn_mi355x.so | cut -c1-16
tail -c 400 $OUT/trace_mid.log
