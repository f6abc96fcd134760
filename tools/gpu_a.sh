#!/bin/bash
# round-2 GPU pass A: full GPU suite + benches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/a
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/a/pytest.log
tail -30 gpurun_out/a/pytest.log
python bench.py > gpurun_out/a/bench_default.log 2> gpurun_out/a/bench_default.err; echo "bench rc=$?"
tail -1 gpurun_out/a/bench_default.log
python bench.py --force-partitioned --L-local 20 --no-cpu-baseline > gpurun_out/a/bench_part20.log 2> gpurun_out/a/bench_part20.err; echo "bench part rc=$?"
tail -1 gpurun_out/a/bench_part20.log
