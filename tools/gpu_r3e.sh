#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/lanczos_small_timing.py > $O/timing.txt 2>&1; cat $O/timing.txt | grep -v amdgpu
