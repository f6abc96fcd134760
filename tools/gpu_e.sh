#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_eig.py -x -q -s > $O/pytest_eig.log 2>&1; echo "pytest eig rc=$?"; tail -12 $O/pytest_eig.log
timeout 300 python tools/bench_arnoldi.py > $O/arnoldi.log 2>&1; cat $O/arnoldi.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/arn -o a -- python3 tools/bench_arnoldi.py > /dev/null 2>&1; S=$(find $O/arn -name "*kernel_stats.csv" | head -1); head -16 "$S" | cut -c1-200; cp "$S" $O/arnoldi_kernel_stats.csv; rm -rf $O/arn
for L in 20 24; do timeout 120 python tools/kbench_spmv.py $L > $O/spmv_L$L.log 2>&1; cat $O/spmv_L$L.log; done
python bench.py --no-cpu-baseline --no-extras --steps 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q > $O/pytest_par.log 2>&1; echo "pytest parity rc=$?"; tail -5 $O/pytest_par.log
