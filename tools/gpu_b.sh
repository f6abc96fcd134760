#!/bin/bash
# round-2 GPU pass B: profiles (kernel stats, PMC traffic), large slabs, full CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/b; mkdir -p $O
python bench.py --no-extras > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.log
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_f.log 2>&1; echo "pmc f rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/pmc_w.log 2>&1; echo "pmc w rc=$?"
find $O -name "*.csv" | head -20
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 > $O/pmc_traffic.log 2>&1; tail -25 $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv
# keep the merged output small: drop raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
python bench.py --force-partitioned --L-local 25 --no-cpu-baseline > $O/bench_part25.log 2> $O/bench_part25.err; echo "part25 rc=$?"; tail -1 $O/bench_part25.log
python bench.py --scaling strong --no-cpu-baseline > $O/bench_strong1.log 2> $O/bench_strong1.err; echo "strong rc=$?"; tail -1 $O/bench_strong1.log; tail -5 $O/bench_strong1.err
python bench.py --steps 2 --warmup 1 --no-extras --no-kernel-events --cpu-full --cpu-threads 8,64 > $O/bench_cpufull.log 2> $O/bench_cpufull.err; echo "cpufull rc=$?"; tail -1 $O/bench_cpufull.log
