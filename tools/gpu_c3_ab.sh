#!/bin/bash
# A/B of library variants (libdsea_<v>.so) on config 3's Lanczos loop: wall clock + per-kernel averages
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp C3_LANCZOS_ONLY=1
for rep in 1 2; do
for v in $1; do
  export DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_$v.so
  echo "== $v: $(python tools/bench_c3.py | head -1)"
  if [ $rep = 2 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_$v -o c3 -- python3 tools/bench_c3.py > /dev/null 2>&1
  head -5 gpurun_out/c3_$v/c3_kernel_stats.csv | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin): print('   %-40s avg %7.2f us' % (r['Name'].split('(')[0][-40:], float(r['AverageNs'])/1e3))"
  fi
done; done
