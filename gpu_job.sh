cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|FAIL|assert" | tail -12
