#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/f; mkdir -p $O
for v in fb1 fb3 fb5 fb7 fb55 ""; do
  lib=dominantsparseeigenad_amd/csrc/libdsea_$v.so; [ -z "$v" ] && lib=dominantsparseeigenad_amd/csrc/libdsea.so
  DSEA_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --steps 5 > $O/bench_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open('$O/bench_$v.log').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], 'spmv', d['roofline']['spmv_avg_launch_ms'])
PY
done
timeout 600 python -m pytest tests/test_gpu_hygiene.py tests/test_gpu_eig.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
