"""BASELINE configs[3] alone (bench.py c4_figures): transfer mat-vec, forward (two Arnoldi solves, k = 200), backward (two
GMRES solves) at D = 512 -- for A/B runs under environment switches (DSEA_WS_SPLIT, DSEA_WS_RPL, DSEA_LIB ...).
    python tools/c4_probe.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("DSEA_C4_SERIAL") == "1":          # left / right solves one after the other instead of on two streams
    import dominantsparseeigenad_amd.eig as _eig
    _eig.CONCURRENT_SIDES = False
dev = torch.device("cuda:0")
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("DSEA_WS_SPLIT", "DSEA_WS_RPL", "DSEA_LIB", "DSEA_TRANSFER_MFMA", "DSEA_C4_SERIAL", "DSEA_ARNOLDI_PIPELINED", "DSEA_ARNOLDI_SPECULATE_PAST_PREDICTION") if k in os.environ)
for _ in range(reps):
    r = bench.c4_figures(dev)
    print("[%s] forward %.2f ms  backward %.2f ms  mat-vec %.2f us  residual %.1e" % (tag or "default", r["forward_ms"], r["backward_ms"],
                                                                                 r["matvec_us"], r["eigen_residual"]))
