#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/pmc2; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC_EA[0-9A-Z_]*|TCC_(HIT|MISS|REQ|READ|TAG_STALL|BUSY)[A-Za-z0-9_]*|TCP_TCC_[A-Z_]*|TCP_PENDING[A-Z_]*|TCP_TA_TCP_STATE_READ|TA_BUSY[a-z_]*|SQ_WAIT[A-Z_]*|SQ_INSTS_VMEM[A-Z_]*|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_ACTIVE_INST[A-Z_]*|SQ_INST_LEVEL_VMEM|TCC_EA0_RD[A-Z0-9_]*)\b" | sort -u > $O/avail.txt
wc -l $O/avail.txt
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python3 tools/pmc_two_passes.py 200 > $O/$name.log 2>&1; f=$(find $O/$name -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "k_rdots" in k or "k_axpy_norm" in k:
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcc2 TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum
find $O -name "*.csv" -size +200k -delete; find $O -name "*.db" -delete
