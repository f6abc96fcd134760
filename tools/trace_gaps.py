"""per-kernel durations and the idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV
    python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]) for r in rows), key=lambda t: t[0])[skip:]
dur, gap_after, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
busy_end = ks[0][0]
total_gap = 0.0
for i, (s, e, name) in enumerate(ks):
    dur[name] += e - s
    cnt[name] += 1
    if i + 1 < len(ks):
        g = ks[i + 1][0] - max(e, busy_end)
        if 0 < g < 200000:            # (gaps above 200 us are host phases -- stage tests -- counted separately)
            gap_after[name] += g
            total_gap += g
    busy_end = max(busy_end, e)
span = ks[-1][1] - ks[0][0]
print("kernels %d  span %.2f ms  sum of durations %.2f ms  short gaps (< 200 us) %.2f ms" % (len(ks), span / 1e6, sum(dur.values()) / 1e6, total_gap / 1e6))
for name in sorted(dur, key=lambda n: -dur[n])[:10]:
    print("  %-48s n %6d  avg %7.2f us   avg gap AFTER it %6.2f us" % (name, cnt[name], dur[name] / cnt[name] / 1e3, gap_after[name] / cnt[name] / 1e3))
