cd $GRAFT_REPO_ROOT
timeout 600 python tools/level1_speed.py 2>&1 | tail -2
