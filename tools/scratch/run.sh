cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/lane_soak.py r50 200 8 608 2>&1 | tail -1
