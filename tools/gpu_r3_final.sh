#!/bin/bash
# round-3 evidence pass at the current commit: full GPU suite, smoke, default bench, rocprofv3 kernel stats + PMC traffic
# of the headline step, library-driver bench + kernel stats at the config-5 per-GPU load, small-problem timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3final; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-anchors --no-live-pmc > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-anchors --no-kernel-events --no-live-pmc > $O/pmc_f.log 2>&1; echo "pmc f rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-anchors --no-kernel-events --no-live-pmc > $O/pmc_w.log 2>&1; echo "pmc w rc=$?"
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 > $O/pmc_traffic.log 2>&1; tail -22 $O/pmc_traffic.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-200
python bench.py --no-cpu-baseline --no-extras --no-anchors --no-live-pmc > $O/bench_after_pmc.log 2>&1; tail -1 $O/bench_after_pmc.log | cut -c1-300
# library driver, one rank over RCCL, config-5 per-GPU load
python bench.py --force-partitioned --L-local 25 --no-cpu-baseline --no-extras > $O/bench_libdriver_2p25.log 2>&1; tail -1 $O/bench_libdriver_2p25.log | cut -c1-500
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats25 -o s -- python3 bench.py --force-partitioned --L-local 25 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events > $O/stats25.log 2>&1; echo "stats25 rc=$?"
S=$(find $O/stats25 -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats_libdriver_2p25.csv; head -14 $O/kernel_stats_libdriver_2p25.csv | cut -c1-200
python bench.py --force-partitioned --no-cpu-baseline --no-extras > $O/bench_libdriver_2p20.log 2>&1; tail -1 $O/bench_libdriver_2p20.log | cut -c1-300
python tools/lanczos_small_timing.py 2>&1 | grep -v amdgpu > $O/lanczos_small.txt; cat $O/lanczos_small.txt
python tools/cg_small_timing.py 2>&1 | grep -v amdgpu > $O/cg_small.txt; cat $O/cg_small.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
