#!/bin/bash
# end-of-round GPU pass: full GPU suite, default bench, rocprofv3 kernel stats + PMC traffic at the current commit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_f.log 2>&1; echo "pmc f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_w.log 2>&1; echo "pmc w rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 > $O/pmc_traffic.log 2>&1; tail -20 $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats.csv; head -10 $O/kernel_stats.csv | cut -c1-220
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
python bench.py --no-cpu-baseline --no-extras > $O/bench2.log 2>&1; tail -1 $O/bench2.log | cut -c1-300
