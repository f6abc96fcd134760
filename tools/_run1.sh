cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 300 python bench.py --gpus 2 --host-staged --steps 1 --warmup 1 > $O/reh2.out 2> $O/reh2.err; echo "rehearsal N=2 rc=$?"; tail -c 1500 $O/reh2.err; tail -1 $O/reh2.out | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_persistent.py -m gpu -q -x -k "mid_size or large" --durations=5 > $O/pytest_mid.log 2>&1; echo "pytest mid rc=$?"; tail -40 $O/pytest_mid.log | cut -c1-300
timeout 300 python tools/lanczos_mid_timing.py 2>&1 | grep -v amdgpu | tee $O/lanczos_mid_timing.txt
timeout 200 python tools/cg_small_timing.py 2>&1 | grep -v amdgpu | tee $O/cg_small.txt
