cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
timeout 300 python bench.py --workload dag --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dag', d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['sweeps_per_step'])"
timeout 300 python bench.py --workload dag --steps 20 --warmup 3 --no-cpu --eps 1e-6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dag', d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['sweeps_per_step'])"
timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grid', d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['sweeps_per_step'])"
