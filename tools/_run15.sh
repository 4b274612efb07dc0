cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_nr_gpu.py -x -q -k headline 2>&1 | grep -v "^E  *+" | tail -15
