#!/bin/bash
# r04 t: the split's last tile as peeled tail code (no wasted QK^T MFMAs): A/B in one process + the GPU tests
mkdir -p gpurun_out/r04t
cd /root/repo
export TMPDIR=/tmp
ab() { timeout 300 python tools/ab_forward.py --config $1 --batch $2 --variants u16 u16+PDSC_ATT_PEEL=0 --rounds 7 --steps $3 > gpurun_out/r04t/ab_$1_$2.txt 2>&1; echo "ab $1 x$2 rc=$?"; grep median gpurun_out/r04t/ab_$1_$2.txt; }
ab n5000_b32 32 20
ab n5000_b32 4 80
ab n5000_b32 1 150
ab lomatch_n10000_b8 8 20
ab lomatch_n10000_b8 1 80
ab kitti_n5000_b16 2 100
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04t/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r04t/pytest_gpu.txt
