cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
timeout 300 python tools/cg_one_exchange_check.py 2>&1 | grep -v amdgpu | tee $O/cg_one_exchange_check.txt
