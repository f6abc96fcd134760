#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py -m gpu -q -x -k "large" --durations=5 > $O/pytest_cgb.log 2>&1; echo "pytest cgb rc=$?"; tail -25 $O/pytest_cgb.log | cut -c1-250
timeout 300 python tools/cg_small_timing.py 2>&1 | grep -v amdgpu | tee $O/cg_timing.txt
