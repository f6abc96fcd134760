#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel trace): per-kernel totals, and busy/idle time of the GPU.
   python tools/prof_summary.py path/to/results.db [steps]"""
import sqlite3, sys
db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
con = sqlite3.connect(db)
cur = con.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-72s %8s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
for r in rows[:28]:
    print("%-72s %8d %10.3f %9.2f %9.2f %9.2f %6.1f" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
ev = cur.execute("select start, end from kernels order by start").fetchall()
span = (ev[-1][1] - ev[0][0]) / 1e6
busy, gaps, last_end = 0.0, [], ev[0][0]
for s, e in ev:
    if s > last_end:
        gaps.append((s - last_end) / 1e3)
    busy += (e - max(s, last_end)) / 1e6 if e > last_end else 0.0
    last_end = max(last_end, e)
gaps.sort()
print("kernel time total %.3f ms (%.3f ms/step over %d steps); span %.3f ms; busy %.3f ms" % (tot, tot / steps, steps, span, busy))
small = [g for g in gaps if g < 50]
big = [g for g in gaps if g >= 50]
print("gaps: %d < 50us (sum %.3f ms, median %.2f us) ; %d >= 50us (sum %.3f ms, max %.1f us)" % (
    len(small), sum(small) / 1e3, small[len(small) // 2] if small else 0, len(big), sum(big) / 1e3, max(big) if big else 0))
