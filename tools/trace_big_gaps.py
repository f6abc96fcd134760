"""the idle gaps above a threshold in a rocprofv3 --kernel-trace CSV, each with the kernel before and after it, and the totals
    python tools/trace_big_gaps.py <kernel_trace.csv> [min_gap_us = 100] [skip_fraction = 0.4]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in rows), key=lambda t: t[0])
ks = ks[int(len(ks) * skip):]
busy_end, t0 = ks[0][1], ks[0][0]
big, small, busy = [], 0, 0
for i in range(1, len(ks)):
    s, e, name = ks[i]
    gap = s - busy_end
    if gap >= thr:
        big.append((gap, (busy_end - t0) / 1e6, ks[i - 1][2], name))
    elif gap > 0:
        small += gap
    busy += max(0, e - max(s, busy_end))
    busy_end = max(busy_end, e)
span = busy_end - t0
print("kernels %d   span %.2f ms   busy %.2f ms   gaps >= %.0f us: %d totalling %.2f ms   smaller gaps %.2f ms"
      % (len(ks), span / 1e6, busy / 1e6, thr / 1e3, len(big), sum(b[0] for b in big) / 1e6, small / 1e6))
for gap, at, before, after in big:
    print("  at +%8.2f ms   %8.1f us   after %-44s before %s" % (at, gap / 1e3, before, after))
