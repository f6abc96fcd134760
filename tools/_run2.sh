cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r05s
python examples/TFIM/sweep.py --N 20 --k 200 --points 4 --data tests/golden/ref_datas 2>&1 | grep -v amdgpu | tee gpurun_out/r05s/sweep.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05s/_tr -o t -- python3 examples/TFIM/sweep.py --N 20 --k 200 --points 4 --data tests/golden/ref_datas > gpurun_out/r05s/sweep_prof.txt 2>&1
T=$(find gpurun_out/r05s/_tr -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$T" 0 | tee gpurun_out/r05s/sweep_gaps.txt
grep "couplings in" gpurun_out/r05s/sweep_prof.txt
rm -rf gpurun_out/r05s/_tr
