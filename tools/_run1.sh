cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eig.py -q 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q -k "transfer or vumps or eig or arnoldi or gmres" 2>&1 | tail -3
