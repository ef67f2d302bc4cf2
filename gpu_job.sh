cd $GRAFT_REPO_ROOT
python tools/tail_stamps.py 2>&1 | tail -14
