#!/bin/bash
# r03 GPU call B: full test log, census (shipped defaults + exact-fp32 mode), settled bench lines, small-problem layer A/B,
# two-steps-in-flight probe.  Outputs -> gpurun_out/r03_b_*
set -u
TAG=r03_b
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 600 python tools/parity_census.py --batches 0,4 > "$OUT/${TAG}_census.txt" 2>&1
timeout 600 python tools/parity_census.py --only n5000_b32 --batches 32,4 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > "$OUT/${TAG}_census_exact_fp32.txt" 2>&1
timeout 600 python tools/parity_census.py --only n5000_b32 --batches 32,4 --compat-format f32 --layer-gemm f32 > "$OUT/${TAG}_census_f32compat_f32gemm.txt" 2>&1
timeout 300 python bench.py > "$OUT/${TAG}_bench_n5000_b32.log" 2>&1; tail -1 "$OUT/${TAG}_bench_n5000_b32.log" > "$OUT/${TAG}_bench_line_n5000_b32.json"
for B in 4 8 16; do
  timeout 200 python bench.py --global-batch $B --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_${B}pairs.json"
done
timeout 200 python bench.py --config kitti_n5000_b16 --global-batch 2 --no-cpu-baseline --no-check 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_kitti_2pairs.json"
timeout 200 python bench.py --config lomatch_n10000_b8 --global-batch 1 --no-cpu-baseline --no-check 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_lomatch_1pair.json"
timeout 200 python bench.py --config n1000_b1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1.json"
timeout 300 python tools/ab_forward.py --config n1000_b1 --rounds 5 --steps 200 --variants u16 u16+PDSC_LAYER_VARIANT=w > "$OUT/${TAG}_ab_n1000.txt" 2>&1
for B in 1 2 3; do
  timeout 300 python tools/ab_forward.py --config n5000_b32 --batch $B --rounds 5 --steps 100 --variants u16 u16+PDSC_LAYER_VARIANT=w u16+PDSC_LAYER_VARIANT=b >> "$OUT/${TAG}_ab_small.txt" 2>&1
done
timeout 200 python tools/overlap_probe.py --n 5000 --bs 32 --steps 40 > "$OUT/${TAG}_overlap_probe.txt" 2>&1
timeout 200 python tools/overlap_probe.py --n 5000 --bs 4 --steps 200 >> "$OUT/${TAG}_overlap_probe.txt" 2>&1
timeout 200 python tools/overlap_probe.py --n 1000 --bs 1 --steps 500 >> "$OUT/${TAG}_overlap_probe.txt" 2>&1
ls -la "$OUT" | grep r03_b
