#!/bin/bash
# r04 s: waves without a valid query skip the matrix / vector work of the attention launch: A/B in one process + the GPU tests
mkdir -p gpurun_out/r04s
cd /root/repo
export TMPDIR=/tmp
for cfg in n5000_b32 lomatch_n10000_b8 kitti_n5000_b16; do
  timeout 300 python tools/ab_forward.py --config $cfg --variants u16 u16+PDSC_ATT_ALL_WAVES=1 --rounds 7 --steps 20 > gpurun_out/r04s/ab_$cfg.txt 2>&1; echo "ab $cfg rc=$?"
  tail -4 gpurun_out/r04s/ab_$cfg.txt
done
timeout 300 python tools/ab_forward.py --config n5000_b32 --batch 4 --variants u16 u16+PDSC_ATT_ALL_WAVES=1 --rounds 7 --steps 60 > gpurun_out/r04s/ab_n5000_4pairs.txt 2>&1; tail -3 gpurun_out/r04s/ab_n5000_4pairs.txt
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04s/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r04s/pytest_gpu.txt
python bench.py > gpurun_out/r04s/bench_n5000_b32.json 2> gpurun_out/r04s/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/r04s/bench_n5000_b32.json').read().strip().splitlines()[-1]); print(d['value'], d['sustained']['value'], d['single_stream']['value'], d['roofline']['avg_launch_ms'], d['roofline']['executed_frac'], d['power']['mean_w'], d['power']['joule_per_pair'], d['check']['ok'])"
