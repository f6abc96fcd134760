#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pro -o p -- python3 tools/partial_reorth_check.py --big > /dev/null 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pro/p_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last partial run = last ~ 1400 kernels; find the last k_pro_update sequence
idx=[i for i,r in enumerate(rows) if 'k_pro_update' in r['Kernel_Name']]
last=idx[-199:]
seg=rows[last[0]-1:last[-1]+5]
import collections
d=collections.defaultdict(list); g=collections.defaultdict(list)
for a,b in zip(seg[:-1],seg[1:]):
    d[a['Kernel_Name'].split('(')[0][-34:]].append((int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3)
    g[a['Kernel_Name'].split('(')[0][-24:]+' -> '+b['Kernel_Name'].split('(')[0][-24:]].append((int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3)
for k,v in d.items(): print("%-36s n %4d  mean %7.2f  median %7.2f us"%(k,len(v),sum(v)/len(v),sorted(v)[len(v)//2]))
for k,v in g.items(): print("gap %-52s n %4d mean %6.2f"%(k,len(v),sum(v)/len(v)))
print("span per step", (int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3/199)
PY
