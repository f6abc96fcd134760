#!/bin/bash
# where the library driver's step differs from the native loop at 2^20 rows on one rank (per-kernel averages per Lanczos step)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in native libdriver; do
  extra=""; [ $v = libdriver ] && extra="--force-partitioned"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ld20_$v -o s -- python3 bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-anchors --no-live-pmc --no-kernel-events > gpurun_out/ld20_$v.log 2>&1
  echo "== $v: $(tail -1 gpurun_out/ld20_$v.log | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])') ms per fwd+bwd under the profiler"
  python3 - $v <<'PY'
import csv,sys
rows=list(csv.DictReader(open('gpurun_out/ld20_%s/s_kernel_stats.csv'%sys.argv[1])))
steps=4*199.0
tot=0
for r in rows[:14]:
    per=float(r['TotalDurationNs'])/1e3/steps
    tot+=per
    print("   %-46s calls %5s avg %8.2f us   per Lanczos step %7.2f us"%(r['Name'].split('(')[0][-46:],r['Calls'],float(r['AverageNs'])/1e3,per))
PY
done
