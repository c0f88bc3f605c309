for r in 1 2 3; do
for d in 1 0; do
BN_RESIDENT_DIRECT=$d python bench.py --no-cpu --no-extras --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('direct=$d', d['ms_per_step'], d['value'])"
done; done
