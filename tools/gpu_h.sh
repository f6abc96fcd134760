#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_eig.py tests/test_gpu_hygiene.py tests/test_gpu_examples.py tests/test_gpu_reference_twins.py tests/test_gpu_bench_contract.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 300 python tools/bench_vumps.py 512 200 > $O/vumps512.log 2>&1; grep -v amdgpu $O/vumps512.log
timeout 300 python tools/bench_vumps.py 100 200 > $O/vumps100.log 2>&1; grep -v "amdgpu\|Warning\|Consider\|  s =" $O/vumps100.log
timeout 600 python examples/TFIM/sweep.py --N 20 --k 200 --data tests/golden/ref_datas --points 20 > $O/sweep_cold.log 2>&1; tail -2 $O/sweep_cold.log
timeout 600 python examples/TFIM/sweep.py --N 20 --k 200 --warm 80 --data tests/golden/ref_datas --points 20 > $O/sweep_warm.log 2>&1; tail -2 $O/sweep_warm.log
