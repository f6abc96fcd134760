#!/bin/bash
# A/B/... of several builds of the library on the SAME box, alternating processes (the dots pass is bimodal per
# process, so every variant is sampled several times):  bash tools/ab.sh "R0 R1 R2" [reps] [bench args...]
# (variants are dominantsparseeigenad_amd/csrc/libdsea_<name>.so; "-" = the in-tree libdsea.so)
cd "$GRAFT_REPO_ROOT" || exit 1
VARS=${1:-"A -"}; REPS=${2:-3}; shift; shift
O=gpurun_out/ab; mkdir -p $O
for rep in $(seq 1 $REPS); do
  for v in $VARS; do
    if [ "$v" = "-" ]; then unset DSEA_LIB; else export DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_$v.so; fi
    python bench.py --no-cpu-baseline --no-extras "$@" > $O/$v$rep.log 2>$O/$v$rep.err
    python - "$v$rep" $O/$v$rep.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d.get("roofline",{})
print(sys.argv[1], "ms/step %.3f"%d["ms_per_step"], "rdots %.2f us"%(r.get("avg_launch_ms",0)*1e3), "axpy %.2f us"%(r.get("other",{}).get("k_axpy_norm",{}).get("avg_launch_ms",0)*1e3), "spmv %.2f us"%(r.get("spmv_avg_launch_ms",0)*1e3))
PY
  done
done
