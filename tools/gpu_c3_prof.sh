#!/bin/bash
# per-kernel durations of config 3's Lanczos loop, automatic geometry vs forced W = 8 / one sub-tile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp C3_LANCZOS_ONLY=1
for v in auto 8; do
  if [ $v = auto ]; then unset C3_SPLIT; else export C3_SPLIT=$v; fi
  python tools/bench_c3.py | head -1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_$v -o c3 -- python3 tools/bench_c3.py > /dev/null 2>&1
  f=$(find gpurun_out/c3_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -8 "$f" | cut -d, -f1-4 | sed 's/(.*)//' | cut -c1-150
done
