#!/bin/bash
# evidence pass for the partial re-orthogonalisation option (default threshold): profiles/r03_partial_reorth.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pro_ev; mkdir -p $O
python tools/partial_reorth_check.py 2>&1 | grep "TFIM\|stencil" > $O/check.txt
python tools/partial_reorth_check.py --big 2>&1 | grep TFIM >> $O/check.txt
python tools/partial_reorth_check.py --L 25 --k 200 2>&1 | grep TFIM >> $O/check.txt
python tools/partial_reorth_check.py --L 28 --k 100 2>&1 | grep TFIM >> $O/check.txt
python tools/pro_delta_sweep.py --L 20 2>&1 | grep delta > $O/delta.txt
python tools/partial_reorth_partitioned.py --L 20 2>&1 | grep "row-part" > $O/part.txt
python tools/partial_reorth_partitioned.py --L 25 2>&1 | grep "row-part" >> $O/part.txt
for r in full partial; do python examples/TFIM/sweep.py --N 20 --k 200 --points 20 --data tests/golden/ref_datas --reorth $r 2>&1 | grep "N=20\|max rel"; done > $O/sweep.txt
for s in 0 1 2 3; do python tools/fuzz_partial_reorth.py --cases 300 --seed $s 2>&1 | grep "MISMATCH\|ERROR\|cases"; done > $O/fuzz.txt
python bench.py --no-cpu-baseline --no-anchors --no-live-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench.py config.partial_reorth_lanczos:', json.dumps(d['config']['partial_reorth_lanczos'])); print('bench.py headline ms_per_step (full schedule):', d['ms_per_step'])" > $O/bench.txt
cat $O/*.txt
