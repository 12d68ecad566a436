#!/bin/bash
# r03 GPU call C: full test log after the in-flight / sampling / small-problem changes, bench lines, in-flight depth A/B.
set -u
TAG=r03_c
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -120 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 300 python bench.py > "$OUT/${TAG}_bench_n5000_b32.log" 2>&1; tail -1 "$OUT/${TAG}_bench_n5000_b32.log" > "$OUT/${TAG}_bench_line_n5000_b32.json"
for D in 1 3; do
  timeout 200 python bench.py --in-flight $D --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_b32_inflight$D.json"
done
for B in 4 8 16; do
  timeout 200 python bench.py --global-batch $B --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_${B}pairs.json"
done
timeout 200 python bench.py --global-batch 4 --in-flight 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_4pairs_inflight3.json"
timeout 200 python bench.py --config kitti_n5000_b16 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_kitti_n5000_b16.json"
timeout 200 python bench.py --config kitti_n5000_b16 --global-batch 2 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_kitti_2pairs.json"
timeout 200 python bench.py --config lomatch_n10000_b8 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_lomatch_n10000_b8.json"
timeout 200 python bench.py --config lomatch_n10000_b8 --global-batch 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_lomatch_1pair.json"
timeout 200 python bench.py --config n1000_b1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1.json"
timeout 200 python bench.py --config n1000_b1 --in-flight 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1_inflight3.json"
timeout 200 python bench.py --config n1000_b1 --in-flight 4 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1_inflight4.json"
ls -la "$OUT" | grep r03_c
