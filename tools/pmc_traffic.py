#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
prescribes) into HBM bytes per launch of each dsea kernel (profiles/r<NN>_pmc_traffic.json format).

Units / corrections (MI355X_MICROARCH.md, section HBM):
  * FETCH_SIZE, WRITE_SIZE are in KiB  -> x 1024;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read -> x 2
    (all dsea kernels read with 16-byte-per-lane coalesced loads); WRITE_SIZE is used uncorrected.
   python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [steps profiled] [out.json]
The result goes to out.json (default gpurun_out/pmc_traffic.json); the copy to keep is committed as
profiles/r<round>_pmc_traffic.json -- bench.py quotes the newest of those when it cannot measure live.
"""
import csv, json, os, sys, collections

def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "").replace("dsea::", "")
            short = short.split("<")[0]
            acc[short][0] += 1
            acc[short][1] += float(row["Counter_Value"])
    return acc

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                  "`bench.py --steps 1 --warmup 1`; bytes = FETCH_SIZE[KiB]*1024*2 (gfx950 wide-read correction) "
                  "+ WRITE_SIZE[KiB]*1024; averaged over all launches of the kernel (i = 1..k-1)"}
for name in sorted(fetch):
    if not name.startswith("k_"):
        continue
    n, fs = fetch[name]
    nw, wsz = write.get(name, [0, 0.0])
    rd = fs / n * 1024 * 2
    wr = (wsz / nw * 1024) if nw else 0.0
    out[name] = {"launches": n, "fetch_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                 "hbm_bytes_per_launch": rd + wr}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# whole-step traffic: the profiled command runs STEPS forward+backward passes (warmup included)
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
out["_steps_profiled"] = STEPS
out["_total_hbm_bytes_per_step"] = sum(v["launches"] * v["hbm_bytes_per_launch"] for kname, v in out.items()
                                       if kname.startswith("k_")) / STEPS
try:
    import subprocess
    out["_commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True,
                                    text=True).stdout.strip() or os.environ.get("DSEA_COMMIT", "unknown")
except Exception:
    out["_commit"] = os.environ.get("DSEA_COMMIT", "unknown")
dest = sys.argv[4] if len(sys.argv) > 4 else os.path.join(root, "gpurun_out", "pmc_traffic.json")
os.makedirs(os.path.dirname(os.path.abspath(dest)), exist_ok=True)
with open(dest, "w") as f:
    json.dump(out, f, indent=1)
for k, v in out.items():
    if not k.startswith("_"):
        print("%-24s launches %5d  read %10.1f MB  write %8.1f MB  total %10.1f MB" % (
            k, v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
