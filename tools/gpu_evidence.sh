#!/bin/bash
# large-slab evidence with the final kernels: config-5 per-GPU load through the partitioned driver, L=28 on one GPU, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/evidence; mkdir -p $O
python bench.py --force-partitioned --L-local 25 --no-cpu-baseline > $O/part25.log 2> $O/part25.err; echo "part25 rc=$?"; tail -1 $O/part25.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p25 -o s -- python3 bench.py --force-partitioned --L-local 25 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/p25stats.log 2>&1; S=$(find $O/p25 -name "*kernel_stats.csv" | head -1); cp "$S" $O/part25_kernel_stats.csv; rm -rf $O/p25; head -8 $O/part25_kernel_stats.csv | cut -c1-200
python bench.py --scaling strong --no-cpu-baseline > $O/strong1.log 2> $O/strong1.err; echo "strong rc=$?"; tail -1 $O/strong1.log | cut -c1-400
python bench.py --scaling strong --reorth none --k 200 --no-cpu-baseline > $O/strong1_bf.log 2> $O/strong1_bf.err; echo "strong bf rc=$?"; tail -1 $O/strong1_bf.log | cut -c1-500
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o s -- python3 tools/bench_c3.py > /dev/null 2>&1; S=$(find $O/c3 -name "*kernel_stats.csv" | head -1); cp "$S" $O/c3_kernel_stats.csv; rm -rf $O/c3
