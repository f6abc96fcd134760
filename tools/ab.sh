#!/bin/bash
# A/B of two builds of the library on the SAME box: libdsea_A.so (baseline) vs libdsea.so, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_A.so; else unset DSEA_LIB; fi
    python bench.py --no-cpu-baseline --no-extras "$@" > $O/$v$rep.log 2>$O/$v$rep.err
    python - "$v$rep" $O/$v$rep.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(sys.argv[1], "ms/step %.3f"%d["ms_per_step"], "rdots %.2f us"%(r.get("avg_launch_ms",0)*1e3), "axpy %.2f us"%(r.get("other",{}).get("k_axpy_norm",{}).get("avg_launch_ms",0)*1e3), "spmv %.2f us"%(r.get("spmv_avg_launch_ms",0)*1e3))
PY
  done
done
