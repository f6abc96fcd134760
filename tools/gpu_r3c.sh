#!/bin/bash
# round 3: library-side partitioned driver -- its tests, then the whole GPU suite, then driver-overhead benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_config5.py -m gpu -q -x --durations=8 -s > $O/pytest_part.log 2>&1; echo "pytest part rc=$?"; tail -30 $O/pytest_part.log
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
# host overhead of the distributed driver at 2^20 rows, world 1 over RCCL: library driver vs Python driver vs native loop
python bench.py --no-cpu-baseline --no-extras --no-anchors > $O/bench_native.log 2>&1; tail -1 $O/bench_native.log | cut -c1-400
python bench.py --force-partitioned --no-cpu-baseline --no-extras > $O/bench_libdriver.log 2>&1; tail -1 $O/bench_libdriver.log | cut -c1-400
DSEA_DRIVER=python python bench.py --force-partitioned --no-cpu-baseline --no-extras > $O/bench_pydriver.log 2>&1; tail -1 $O/bench_pydriver.log | cut -c1-400
