#!/bin/bash
# r03 GPU call A: validate the new unorm16 compat kernel + small-launch layer shapes, census with the shipped defaults,
# kernel stats of the per-GPU shares, A/B of the experiments knobs.  Outputs -> gpurun_out/r03_a_*
set -u
TAG=r03_a
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
EXP=$ROOT/pointdsc_amd/libpointdsc_hip_exp.so
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 200 python tools/compat_bench.py --exp > "$OUT/${TAG}_compat_bench.txt" 2>&1
timeout 600 python tools/parity_census.py --batches 0,1,2,4,8,16 > "$OUT/${TAG}_census.txt" 2>&1
timeout 300 python bench.py > "$OUT/${TAG}_bench_n5000_b32.log" 2>&1; tail -1 "$OUT/${TAG}_bench_n5000_b32.log" > "$OUT/${TAG}_bench_line_n5000_b32.json"
for B in 4 8 16; do
  timeout 200 python bench.py --global-batch $B --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_${B}pairs.json"
done
timeout 200 python bench.py --config kitti_n5000_b16 --global-batch 2 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_kitti_2pairs.json"
timeout 200 python bench.py --config lomatch_n10000_b8 --global-batch 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_lomatch_1pair.json"
timeout 200 python bench.py --config n1000_b1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1.json"
# A/B (experiments library)
timeout 300 python tools/ab_forward.py --config n5000_b32 --rounds 5 --steps 15 --variants u16 u16+PDSC_COMPAT16_VARIANT=0 u16+PDSC_ATT_SPLIT_NW=4 u16+PDSC_COMPAT16_VARIANT=2 > "$OUT/${TAG}_ab_b32.txt" 2>&1
timeout 300 python tools/ab_forward.py --config n5000_b32 --batch 4 --rounds 5 --steps 60 --variants u16 u16+PDSC_LAYER_H3_SHAPE=42 u16+PDSC_LAYER_H3_SHAPE=22 u16+PDSC_LAYER_H3_SHAPE=13 u16+PDSC_LAYER_H3_SHAPE=24 u16+PDSC_LAYER_H3_SHAPE=14 u16+PDSC_ATT_SPLIT_NW=4 u16+PDSC_LAYER_VARIANT=b > "$OUT/${TAG}_ab_b4.txt" 2>&1
timeout 300 python tools/ab_forward.py --config n5000_b32 --batch 8 --rounds 5 --steps 40 --variants u16 u16+PDSC_LAYER_H3_SHAPE=42 u16+PDSC_LAYER_H3_SHAPE=22 u16+PDSC_LAYER_H3_SHAPE=24 > "$OUT/${TAG}_ab_b8.txt" 2>&1
timeout 300 python tools/ab_forward.py --config kitti_n5000_b16 --batch 2 --rounds 5 --steps 80 --variants u16 u16+PDSC_LAYER_H3_SHAPE=42 u16+PDSC_LAYER_H3_SHAPE=23 u16+PDSC_LAYER_H3_SHAPE=14 u16+PDSC_LAYER_VARIANT=b > "$OUT/${TAG}_ab_kitti2.txt" 2>&1
timeout 300 python tools/ab_forward.py --config lomatch_n10000_b8 --batch 1 --rounds 5 --steps 60 --variants u16 u16+PDSC_LAYER_H3_SHAPE=42 u16+PDSC_LAYER_H3_SHAPE=23 u16+PDSC_LAYER_H3_SHAPE=14 u16+PDSC_LAYER_VARIANT=b > "$OUT/${TAG}_ab_lomatch1.txt" 2>&1
# kernel stats of the shares
cd /tmp
for spec in "n5000_b32 4" "kitti_n5000_b16 2" "lomatch_n10000_b8 1" "n1000_b1 1" "n5000_b32 32"; do
  set -- $spec; c=$1; b=$2
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o k -- python "$ROOT/bench.py" --config $c --global-batch $b --steps 6 --warmup 1 --no-cpu-baseline --no-check --sustain-seconds 0 > "$OUT/${TAG}_rocprof_${c}_$b.log" 2>&1
  DB=$(find /tmp/prof_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_kernel_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_${c}_${b}pairs.txt" 2>&1
  rm -rf /tmp/prof_$c
done
ls -la "$OUT" | grep r03_a
