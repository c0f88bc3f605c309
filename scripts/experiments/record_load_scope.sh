# TIMING ONLY: the resident kernel's record loads with sc1 (product), sc0 (served by the XCD's own L2: stale for cross-XCD edges)
# and no scope bit (L1 may serve them: stale).  Libraries: build/dbg_csrc with -DBN_LD_AUX=1 / 0.
for r in 1 2; do
for lib in bayesiannetwork_amd/libbn_mi355x.so build/libbn_sc0.so build/libbn_nosc.so; do
BN_MI355X_LIB=$lib python bench.py --no-cpu --no-extras --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', 'ms', round(d['ms_per_step'],4), 'sweeps', d['config']['sweeps_per_step'], 'us/sweep', round(d['roofline'].get('avg_sweep_us_devclock'),2))"
done; done
