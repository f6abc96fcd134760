#!/bin/bash
# ONE parametrised evidence script for the GPU box (replaces the per-session run scripts of rounds 1-3):
#
#     gpurun --timeout 1800 -- 'bash tools/gpu_evidence.sh <recipe> [<recipe> ...]'
#
# Every recipe writes under gpurun_out/$TAG/ (TAG defaults to "ev"; the summaries worth keeping are copied to
# profiles/ by hand, named r<round>_*).  Recipes with arguments take them as recipe:arg1:arg2 (":" separated).
#
#   tests[:k-expr]      pytest -m gpu (optionally -k <expr>)             smoke          __graft_entry__.smoke()
#   bench               default bench.py line                            benchq         bench without baseline / anchors / pmc
#   stats               rocprofv3 --kernel-trace --stats of 3 headline steps -> kernel_stats.csv
#   pmc                 FETCH_SIZE / WRITE_SIZE passes (separate) -> pmc_traffic.json via tools/pmc_traffic.py
#   callable            kernel traces of tools/bench_generic.py (native / lambda / gather-table operand) through tools/trace_gaps.py
#   pmcsell             the same with --operator sell -> pmc_traffic_sell.json          statsop:OP   kernel stats with --operator OP
#   libdriver:LL        bench --force-partitioned --L-local LL (library driver, one rank over RCCL) + its kernel stats
#   ldprof[:LL]         native loop vs library driver at 2^LL rows (default 20), per-kernel averages per Lanczos step
#   c3                  config 3 (stencil N = 1e5, k = 300): timings + kernel stats (tools/bench_c3.py)
#   small               single-launch regimes: tools/lanczos_small_timing.py, tools/cg_small_timing.py
#   anchors             bench.py one-GPU anchors (L = 28 k = 100 / k = 80 shadow on+off; 2^25 rows k = 200)
#   rehearsal           bench.py --host-staged at N = 2, 4, 8 (the real N > 1 branch, ranks sharing the GPU, toy sizes)
#   rehearsal_rccl      the same with the library's RCCL branch executing over the stand-in RCCL (DSEA_RCCL_LIB)
#   watchdog:KIND[:S]   N = 2 rehearsal with a hang injected inside the stand-in (exchange | allreduce), stall deadline S s
#   fuzzpart:SEEDS:N    tools/fuzz_partitioned.py for each seed (comma separated), N cases each
#   fuzz:SEEDS:N        tools/fuzz_parity.py likewise
#   ab:VARIANTS:REPS    alternate library builds csrc/libdsea_<name>.so ("-" = in-tree) on the headline bench
#   abenv:VAR:VALS:REPS alternate an environment switch (VALS comma separated)
#   py:SCRIPT[:ARGS]    python SCRIPT ARGS ("," -> " ") with stdout kept
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
TAG=${TAG:-ev}
O=gpurun_out/$TAG; mkdir -p "$O"
Q="--no-cpu-baseline --no-extras --no-anchors --no-live-pmc"

line() { tail -1 "$1" | cut -c1-${2:-400}; }
stats_of() {  # stats_of <outname> <bench args...>: rocprofv3 kernel stats of a bench invocation -> $O/<outname>.csv
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_st_$name" -o s -- python3 bench.py "$@" > "$O/$name.statslog" 2>&1
  echo "stats $name rc=$?"
  local S; S=$(find "$O/_st_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp "$S" "$O/$name.csv" && head -12 "$O/$name.csv" | cut -c1-200
  rm -rf "$O/_st_$name"
}
summarise() { python3 - "$@" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "rdots %.2f us" % (r.get("avg_launch_ms", 0) * 1e3),
      "axpy %.2f us" % (r.get("other", {}).get("k_axpy_norm", {}).get("avg_launch_ms", 0) * 1e3),
      "spmv %.2f us" % (r.get("spmv_avg_launch_ms", 0) * 1e3), d["config"].get("basis_placement_probe_us"))
PY
}

for spec in "$@"; do
  IFS=: read -r recipe a1 a2 a3 <<< "$spec"
  echo "=== $spec"
  case $recipe in
    tests)
      if [ -n "$a1" ]; then timeout 3000 python -m pytest tests -m gpu -q --durations=8 -k "$a1" > "$O/pytest.log" 2>&1
      else timeout 3000 python -m pytest tests -m gpu -q --durations=8 > "$O/pytest.log" 2>&1; fi
      echo "pytest rc=$?"; tail -15 "$O/pytest.log" | cut -c1-250 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$O/smoke.log" ;;
    bench) python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; line "$O/bench.json" 600 ;;
    benchq) python bench.py $Q > "$O/benchq.json" 2> "$O/benchq.err"; echo "rc=$?"; line "$O/benchq.json" ;;
    stats) stats_of kernel_stats --steps 3 --warmup 1 $Q ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/_pmc_$c" -o p -- python3 bench.py --steps 1 --warmup 1 $Q --no-kernel-events > "$O/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
      done
      F=$(find "$O/_pmc_FETCH_SIZE" -name "*counter_collection.csv" | head -1); W=$(find "$O/_pmc_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
      DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 "$O/pmc_traffic.json" > "$O/pmc_traffic.log" 2>&1; tail -24 "$O/pmc_traffic.log"
      rm -rf "$O"/_pmc_* ;;
    pmcsell)   # the same two passes with the operator as an explicit SELL matrix (raw FETCH_SIZE of k_spmv_sell: its 8 / 2-byte
               # per-lane loads are not the 16-byte streaming reads the x2 correction was calibrated on -- read it both ways)
      for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/_pmcs_$c" -o p -- python3 bench.py --operator sell --steps 1 --warmup 1 $Q --no-kernel-events > "$O/pmcsell_$c.log" 2>&1; echo "pmc $c rc=$?"
      done
      F=$(find "$O/_pmcs_FETCH_SIZE" -name "*counter_collection.csv" | head -1); W=$(find "$O/_pmcs_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
      DSEA_COMMIT=$(cat .commit 2>/dev/null) python tools/pmc_traffic.py "$F" "$W" 2 "$O/pmc_traffic_sell.json" > "$O/pmc_traffic_sell.log" 2>&1; tail -24 "$O/pmc_traffic_sell.log"
      rm -rf "$O"/_pmcs_* ;;
    callable)  # the reference's calling convention (A an opaque callable): kernel trace per mode -> durations and idle gaps
      for m in native callable tables; do
        rocprofv3 --kernel-trace --output-format csv -d "$O/_tr_$m" -o t -- python3 tools/bench_generic.py --mode $m --reps 4 > "$O/callable_$m.log" 2>&1
        T=$(find "$O/_tr_$m" -name "*kernel_trace.csv" | head -1)
        echo "== $m: $(grep 'fwd+bwd' "$O/callable_$m.log" | cut -c1-200)"
        python3 tools/trace_gaps.py "$T" 0 | head -9
        rm -rf "$O/_tr_$m"
      done ;;
    statsop)   # statsop:<operator>: rocprofv3 kernel stats of 3 headline steps with --operator <operator>
      stats_of kernel_stats_$a1 --operator $a1 --steps 3 --warmup 1 $Q ;;
    libdriver)
      LL=${a1:-20}
      python bench.py --force-partitioned --L-local $LL --no-cpu-baseline --no-extras > "$O/libdriver_2p$LL.json" 2> "$O/libdriver_2p$LL.err"; echo "rc=$?"; line "$O/libdriver_2p$LL.json" 500
      stats_of libdriver_2p${LL}_kernel_stats --force-partitioned --L-local $LL --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events ;;
    ldprof)
      LL=${a1:-20}; NS=3; [ "$LL" -ge 24 ] && NS=1
      for v in native libdriver; do
        extra=""; [ $v = libdriver ] && extra="--force-partitioned"
        stats_of ld${LL}_$v $extra --L-local $LL --k 200 --steps $NS --warmup 1 $Q --no-kernel-events > /dev/null
        echo "== $v: $(tail -1 "$O/ld${LL}_$v.statslog" | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])') ms per fwd+bwd under the profiler"
        python3 - "$O/ld${LL}_$v.csv" $NS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = (int(sys.argv[2]) + 1) * 199.0
for r in rows[:16]:
    per = float(r['TotalDurationNs']) / 1e3 / steps
    print("   %-46s calls %5s avg %8.2f us   per Lanczos step %7.2f us" % (r['Name'].split('(')[0][-46:], r['Calls'], float(r['AverageNs']) / 1e3, per))
PY
      done ;;
    c3)
      python tools/bench_c3.py 2>&1 | grep -v amdgpu | tee "$O/c3.txt"
      rocprofv3 --kernel-trace --stats --output-format csv -d "$O/_c3" -o c3 -- python3 tools/bench_c3.py > /dev/null 2>&1
      S=$(find "$O/_c3" -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" "$O/c3_kernel_stats.csv" && head -10 "$O/c3_kernel_stats.csv" | cut -c1-200; rm -rf "$O/_c3" ;;
    small)
      python tools/lanczos_small_timing.py 2>&1 | grep -v amdgpu | tee "$O/lanczos_small.txt"
      python tools/cg_small_timing.py 2>&1 | grep -v amdgpu | tee "$O/cg_small.txt" ;;
    anchors)
      python bench.py --anchors-only > "$O/anchors.json" 2> "$O/anchors.err"; echo "anchors rc=$?"; line "$O/anchors.json" 1200 ;;
    rehearsal)
      for n in 2 4 8; do
        timeout 900 python bench.py --gpus $n --host-staged --steps 2 --warmup 1 > "$O/rehearsal_n$n.json" 2> "$O/rehearsal_n$n.err"; echo "rehearsal N=$n rc=$?"; line "$O/rehearsal_n$n.json" 700
      done ;;
    rehearsal_rccl)   # the same rehearsal with the LIBRARY'S OWN RCCL calls executing over the stand-in of tests/fake_rccl
      for n in 2 4 8; do
        DSEA_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 900 python bench.py --gpus $n --host-staged --steps 2 --warmup 1 > "$O/rehearsal_rccl_n$n.json" 2> "$O/rehearsal_rccl_n$n.err"; echo "rehearsal (stand-in RCCL) N=$n rc=$?"; line "$O/rehearsal_rccl_n$n.json" 300
      done ;;
    watchdog)         # watchdog:exchange / watchdog:allreduce -- a hang injected INSIDE the stand-in RCCL, N = 2
      DSEA_BENCH_INJECT_HANG=$a1 DSEA_BENCH_STALL_S=${a2:-20} DSEA_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 900 python bench.py --gpus 2 --host-staged --steps 2 --warmup 1 > "$O/watchdog_$a1.json" 2> "$O/watchdog_$a1.err"; echo "watchdog $a1 rc=$?"
      tail -1 "$O/watchdog_$a1.json" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print("stage", c["fallback_stage"], "|", c["fallback_stage_is"], "|", c["fallback_reason"], "|", c.get("partitioned_driver"))' ;;
    fuzzpart)
      for s in ${a1//,/ }; do python tools/fuzz_partitioned.py --cases ${a2:-30} --seed $s 2>&1 | grep -v "amdgpu\|Gloo" > "$O/fuzz_partitioned_seed$s.txt"; tail -1 "$O/fuzz_partitioned_seed$s.txt"; done ;;
    fuzz)
      for s in ${a1//,/ }; do python tools/fuzz_parity.py --cases ${a2:-300} --seed $s 2>&1 | grep -v amdgpu > "$O/fuzz_parity_seed$s.txt"; tail -2 "$O/fuzz_parity_seed$s.txt"; done ;;
    ab)
      for rep in $(seq 1 ${a2:-3}); do for v in ${a1//,/ }; do
        if [ "$v" = "-" ]; then unset DSEA_LIB; else export DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_$v.so; fi
        python bench.py $Q > "$O/ab_$v$rep.json" 2> "$O/ab_$v$rep.err"; summarise "$v#$rep" "$O/ab_$v$rep.json"
      done; done; unset DSEA_LIB ;;
    abenv)
      for rep in $(seq 1 ${a3:-3}); do for v in ${a2//,/ }; do
        env $a1=$v python bench.py $Q > "$O/abenv_$v$rep.json" 2> "$O/abenv_$v$rep.err"; summarise "$a1=$v#$rep" "$O/abenv_$v$rep.json"
      done; done ;;
    py)
      name=$(basename "$a1" .py)
      python "$a1" ${a2//,/ } 2>&1 | grep -v amdgpu | tee "$O/$name${a3:+_$a3}.txt" | tail -60 ;;
    *) echo "unknown recipe $recipe"; exit 2 ;;
  esac
done
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*.db" -delete; find "$O" -name "*agent_info.csv" -delete
exit 0
