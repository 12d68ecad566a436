#!/bin/bash
# r04 q: whole-forward power record + layer-launch round quantisation (time per tile at 2.0 / 2.45 / 3.0 / 3.45 / 4.0 rounds)
mkdir -p gpurun_out/r04q
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/forward_power.py --seconds 5 > gpurun_out/r04q/forward_power.txt 2> gpurun_out/r04q/forward_power.err; echo "forward power rc=$?"
timeout 200 python tools/forward_power.py --seconds 4 --pairs 4 > gpurun_out/r04q/forward_power_4pairs.txt 2>> gpurun_out/r04q/forward_power.err; echo "forward power 4 rc=$?"
for bs in 20 26 32 39 45 52; do
  timeout 200 python tools/layer_bench.py --pf --bs $bs --rounds 4 --calls 20 --variants PDSC_LAYER_GEMM=1,PDSC_LAYER_H3_COOP=0 > gpurun_out/r04q/layer_bs$bs.txt 2>&1; echo "layer bs=$bs rc=$?"
done
tail -n 3 gpurun_out/r04q/layer_bs*.txt
cat gpurun_out/r04q/forward_power.txt gpurun_out/r04q/forward_power_4pairs.txt
tail -5 gpurun_out/r04q/forward_power.err
