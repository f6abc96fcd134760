#!/bin/bash
# round-2 GPU pass D: f-1 in the library, persistent CG after the heuristics change
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_eig.py -x -q -s > $O/pytest_eig.log 2>&1; echo "pytest eig rc=$?"; tail -25 $O/pytest_eig.log
timeout 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_hygiene.py tests/test_gpu_examples.py tests/test_gpu_reference_twins.py -x -q > $O/pytest_misc.log 2>&1; echo "pytest misc rc=$?"; tail -8 $O/pytest_misc.log
timeout 300 python tools/bench_vumps.py 512 200 > $O/vumps512.log 2>&1; cat $O/vumps512.log
DSEA_VUMPS_CALLABLE=1 timeout 300 python tools/bench_vumps.py 512 200 > $O/vumps512_callable.log 2>&1; cat $O/vumps512_callable.log
timeout 300 python tools/bench_vumps.py 100 200 > $O/vumps100.log 2>&1; cat $O/vumps100.log
timeout 300 python tools/bench_c3.py > $O/c3.log 2>&1; cat $O/c3.log
