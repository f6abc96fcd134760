cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_reference_twins.py -m gpu -q -x -k "mid_size or config3" --durations=3 > $O/pytest_mid.log 2>&1; echo "pytest mid rc=$?"; tail -8 $O/pytest_mid.log | cut -c1-300
DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_TIM.so timeout 200 python tools/lanczos_mid_phase_timing.py 2>&1 | grep -v amdgpu | tee $O/lanczos_mid_phases.txt
timeout 300 python tools/lanczos_mid_timing.py 2>&1 | grep -v amdgpu | tee $O/lanczos_mid_timing.txt
