cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_0_world8.py tests/test_gpu_bench_contract.py tests/test_gpu_config5.py tests/test_gpu_partitioned.py -m gpu -q -rs --durations=5 -k "world8 or rehearsal or config5 or anchor or rebinding" > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -16 $O/pytest_a.log | cut -c1-250
ps aux | grep -i python | grep -v grep | wc -l; rocm-smi --showpids 2>/dev/null | head -20
