# preferred number of workgroups of the mid-size kernel (BN_MID_PARTS), us per sweep / us per query; run on the GPU box
for pp in 16 24 32 48; do echo "== preferred parts $pp"; BN_MID_PARTS=$pp BN_MID=2 timeout 300 python scripts/experiments/mid_path.py mixed80 mixed300 mixed1000 dag60k4 dag200k4 2>&1 | sed -E "s/ aborts.*'mid': /  mid: /" | cut -c1-200; done
