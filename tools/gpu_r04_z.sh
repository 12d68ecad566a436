#!/bin/bash
# r04 z: spectral-matching baseline with the matrix in the register file: bit identity with the streaming form + timing
mkdir -p gpurun_out/r04z
cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sm_baseline or cal_confidence or other_entry_points" > gpurun_out/r04z/pytest_sm.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04z/pytest_sm.txt
timeout 200 python tools/sm_bench.py > gpurun_out/r04z/sm_bench.txt 2>&1; echo "sm_bench rc=$?"; grep -v amdgpu gpurun_out/r04z/sm_bench.txt
