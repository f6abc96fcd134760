#!/bin/bash
# round-2 GPU pass C: new tests (persistent CG, hygiene, partitioned), config-3 tool, profiles with csv output, L=28
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_hygiene.py -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -15 $O/pytest_new.log
timeout 300 python tools/bench_c3.py > $O/c3.log 2>&1; cat $O/c3.log
timeout 300 python tools/bench_c3.py 20000 > $O/c3_20000.log 2>&1; cat $O/c3_20000.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_persistent.py --deselect tests/test_gpu_hygiene.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -8 $O/pytest_all.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_f.log 2>&1; echo "pmc f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_w.log 2>&1; echo "pmc w rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 > $O/pmc_traffic.log 2>&1; tail -25 $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats.csv; head -14 $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
python bench.py --scaling strong --no-cpu-baseline > $O/bench_strong1.log 2> $O/bench_strong1.err; echo "strong rc=$?"; tail -1 $O/bench_strong1.log; tail -5 $O/bench_strong1.err
