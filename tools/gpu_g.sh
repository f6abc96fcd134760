#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/g; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
timeout 200 python tools/kbench_symv.py > $O/symv.log 2>&1; cat $O/symv.log
timeout 200 python tools/bench_c3.py > $O/c3.log 2>&1; cat $O/c3.log
python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log
python bench.py --scaling strong --reorth none --k 200 --no-cpu-baseline > $O/bench_L28_k200_basisfree.log 2> $O/bench_L28.err; tail -1 $O/bench_L28_k200_basisfree.log; tail -3 $O/bench_L28.err
