#!/bin/bash
# round 3, first GPU pass: full GPU suite (new: config-5 slab, L=28 anchor, L=20 tight-eps fixture, hygiene), default
# bench with the live one-GPU anchors, partitioned-path fuzz judged against the single-GPU self-spread (3 seeds x 30)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
grep -E "config-5 slab|L=28 k=100|L=20 k=200 eps" $O/pytest.log
python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-6000
for s in 0 1 2; do
  timeout 1500 python tools/fuzz_partitioned.py --cases 30 --seed $s > $O/fuzz_seed$s.txt 2>$O/fuzz_seed$s.err; echo "fuzz seed $s rc=$?"; tail -1 $O/fuzz_seed$s.txt
done
grep -h "FAIL\|UNEXPLAINED" $O/fuzz_seed*.txt | head -20
