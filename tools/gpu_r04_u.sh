#!/bin/bash
# r04 u: census at the reference's multiway size (16 pairs of N = 20000) + the experiments-build test of the attention savings
mkdir -p gpurun_out/r04u
cd /root/repo
export TMPDIR=/tmp
timeout 600 python tools/parity_census.py --only multiway_n20000_b1 --batches 1,2,4,8,16 > gpurun_out/r04u/census_multiway.txt 2>&1; echo "census rc=$?"
timeout 300 python tools/parity_census.py --only multiway_n20000_b1 --batches 1,2 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > gpurun_out/r04u/census_multiway_exact_fp32.txt 2>&1; echo "census fp32 rc=$?"
cut -c1-400 gpurun_out/r04u/census_multiway.txt | grep -v amdgpu
cut -c1-300 gpurun_out/r04u/census_multiway_exact_fp32.txt | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "parity_census and multiway" > gpurun_out/r04u/pytest_census.txt 2>&1; echo "pytest census rc=$?"; tail -2 gpurun_out/r04u/pytest_census.txt
POINTDSC_HIP_LIB=pointdsc_amd/libpointdsc_hip_exp.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_wasted_work or persistent_attention" > gpurun_out/r04u/pytest_exp.txt 2>&1; echo "pytest exp rc=$?"; tail -3 gpurun_out/r04u/pytest_exp.txt
