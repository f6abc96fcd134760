"""timeline (start offset, duration, gap before) of the kernels between the LAST fused Lanczos tail of a forward pass and the first
dots pass of the next one -- the Ritz step, the backward pass and the host glue of one headline step
    python tools/trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-56:]) for r in rows), key=lambda t: t[0])
tails = [i for i, k in enumerate(ks) if "k_spmv_tfim<11, true>" in k[2]]
# boundaries: a tail followed (within a few kernels) by something that is not k_rdots / finalize / axpy
ends = [i for a, i in enumerate(tails) if a + 1 == len(tails) or tails[a + 1] - i > 8]
if len(ends) < 2:
    sys.exit("no complete step in the trace")
i0 = ends[-2]
i1 = next(j for j in range(i0 + 1, len(ks)) if "k_rdots" in ks[j][2])
t0 = ks[i0][1]
print("between the end of a forward's Lanczos loop and the first dots pass of the next step: %.3f ms, %d kernels" % ((ks[i1][0] - t0) / 1e6, i1 - i0 - 1))
busy = 0
for j in range(i0 + 1, i1 + 1):
    s, e, name = ks[j]
    gap = s - ks[j - 1][1]
    busy += e - s
    print("  +%8.1f us  gap %7.1f us  dur %8.1f us  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, name))
print("busy %.3f ms of it" % ((busy - (ks[i1][1] - ks[i1][0])) / 1e6))
