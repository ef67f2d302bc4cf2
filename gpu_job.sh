cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1c -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_r1c.log 2>&1; echo "trace rc=$?"; tail -2 gpurun_out/prof_r1c.log
cat $(find gpurun_out/prof_r1c -name "*kernel_stats.csv" | head -1)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_sq/*/ 2>/dev/null | head -20
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_fetch","pmc_write","pmc_sq"):
    for f in glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"][:40]; acc[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
        for k in acc:
            print(tag, k, {c: (v/cnt[(k,c)]) for c,v in acc[k].items()}, "dispatches", max(cnt[(k,c)] for c in acc[k]))
PY
