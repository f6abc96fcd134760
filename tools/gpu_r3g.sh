#!/bin/bash
# reference example workload at its own size (examples/TFIM/E0.py, chiF.py: N = 10, k = 300; 100 couplings, second order):
# round-3 single-launch kernels on and off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
for N in 10 12; do
  echo "== N=$N k=300, single-launch Lanczos + CG (default)"; python examples/TFIM/sweep.py --N $N --k 300 --data tests/golden/ref_datas --points 100 2>&1 | grep -v amdgpu | tail -3
  echo "== N=$N k=300, multi-launch kernels"; DSEA_NO_PERSIST=1 python examples/TFIM/sweep.py --N $N --k 300 --data tests/golden/ref_datas --points 100 2>&1 | grep -v amdgpu | tail -3
done 2>&1 | tee $O/sweep_small.txt
