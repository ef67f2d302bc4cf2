cd $GRAFT_REPO_ROOT
python tools/phase_stamps.py 2>&1 | grep -A14 "B=256"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
