#!/bin/bash
# round 3: single-launch Lanczos -- tests (under a timeout: a persistent kernel must not hang the box) and timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py -m gpu -q -x -k "single_launch" --durations=5 > $O/pytest_lzp.log 2>&1; echo "pytest lzp rc=$?"; tail -25 $O/pytest_lzp.log
timeout 600 python tools/lanczos_small_timing.py > $O/timing.txt 2>&1; echo "timing rc=$?"; cat $O/timing.txt
