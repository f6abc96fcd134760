cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_eig.py -m gpu -q -x --durations=3 > $O/pytest_eig.log 2>&1; echo "pytest eig rc=$?"; tail -12 $O/pytest_eig.log | cut -c1-300
timeout 200 python tools/kbench_transfer.py 2>&1 | grep -v amdgpu | head -3 | tee $O/kbench_transfer.txt
DSEA_TRANSFER_ROCBLAS=1 timeout 200 python tools/kbench_transfer.py 2>&1 | grep -v amdgpu | head -1 | sed 's/^/rocBLAS path: /' | tee -a $O/kbench_transfer.txt
for D in 128 256; do timeout 200 python tools/kbench_transfer.py $D 2>&1 | grep -v amdgpu | head -1 | sed "s/^/D=$D: /" | tee -a $O/kbench_transfer.txt; done
timeout 300 python tools/bench_vumps.py 2>&1 | grep -v amdgpu | tail -6 | tee $O/bench_vumps.txt
