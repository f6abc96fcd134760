#!/bin/bash
# round 3: GPU suite (all failures listed, no -x) + partitioned fuzz at both CG tolerances
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=12 -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log
grep -E "config-5 slab|L=28 k=100|L=20 k=200 eps" $O/pytest.log
for s in 0 1 2; do
  timeout 1500 python tools/fuzz_partitioned.py --cases 30 --seed $s > $O/fuzz_seed$s.txt 2>$O/fuzz_seed$s.err; echo "fuzz seed $s rc=$?"; tail -1 $O/fuzz_seed$s.txt
done
grep -h "FAIL\|UNEXPLAINED" $O/fuzz_seed*.txt | cut -c1-300 | head -20
