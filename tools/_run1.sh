cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_hygiene.py tests/test_gpu_partitioned.py -m gpu -q --durations=14 -k "not rehearsal and (bench or hygiene or test_gpu_hygiene or hip_backend)" > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -22 $O/pytest_a.log | cut -c1-180
ps aux | grep -c python
