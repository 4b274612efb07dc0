cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_guard_gpu.py -x -q 2>&1 | grep -v "^E  *+" | tail -20
