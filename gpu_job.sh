cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bp_gpu.py -x -q 2>&1 | tail -5
timeout 300 python -m pytest tests/test_lw_gpu.py -x -q -k "bit_exact or split or error" 2>&1 | tail -5
python bench.py --steps 30 --warmup 5 --no-cpu 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --eps 1e-6 --no-cpu 2>&1 | tail -1
python bench.py --steps 10 --warmup 3 --rows 2048 --cols 2048 --no-cpu 2>&1 | tail -1
