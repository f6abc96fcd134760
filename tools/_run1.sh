cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=$PWD
for v in "" _D1 _D2; do
cd /tmp; rm -rf /tmp/tp; DSEA_LIB=$R/dominantsparseeigenad_amd/csrc/libdsea$v.so DSEA_TRANSFER_MFMA=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o t -- python3 $R/tools/kbench_transfer.py > /dev/null 2>&1
S=$(find /tmp/tp -name "*kernel_stats.csv" | head -1); echo "== variant '$v'"; grep -i "dgemm_mfma" $S | sed 's/.*k_dgemm_mfma\(<[^>]*>\)[^"]*"/\1/' | cut -c1-100
done
