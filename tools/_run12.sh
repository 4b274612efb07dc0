cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^E  *+" | tail -12
JG_TOP_LEVEL=2 JG_TOP_FRONT=16 timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
