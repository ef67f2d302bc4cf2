cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r1b.json 2> gpurun_out/bench_r1b.err; echo "bench rc=$?"
cat gpurun_out/bench_r1b.json; tail -3 gpurun_out/bench_r1b.err
python bench.py --steps 20 --warmup 5 --max-iter 100 --no-cpu-baseline > gpurun_out/bench_r1b_it100.json 2>&1; cat gpurun_out/bench_r1b_it100.json
