#!/bin/bash
# the same library, alternating an environment switch between processes:  bash tools/ab_env.sh VAR "v1 v2" reps
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; VALS=$2; REPS=${3:-3}
O=gpurun_out/ab; mkdir -p $O
for rep in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-extras > $O/e$v$rep.log 2>$O/e$v$rep.err
    python - "$VAR=$v #$rep" $O/e$v$rep.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(sys.argv[1], "ms/step %.3f"%d["ms_per_step"], "rdots %.2f us"%(r.get("avg_launch_ms",0)*1e3), "axpy %.2f us"%(r.get("other",{}).get("k_axpy_norm",{}).get("avg_launch_ms",0)*1e3), "spmv %.2f us"%(r.get("spmv_avg_launch_ms",0)*1e3), d["config"].get("basis_placement_probe_us"))
PY
  done
done
