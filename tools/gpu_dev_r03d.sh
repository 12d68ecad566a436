#!/bin/bash
# r03 GPU call D: full test log (kitti census fixture present), census of all available families at every batch size, bench lines.
set -u
TAG=r03_d
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 900 python tools/parity_census.py --batches 0,1,2,4,8,16,32 > "$OUT/${TAG}_census.txt" 2>&1
timeout 300 python bench.py > "$OUT/${TAG}_bench_n5000_b32.log" 2>&1; tail -1 "$OUT/${TAG}_bench_n5000_b32.log" > "$OUT/${TAG}_bench_line_n5000_b32.json"
timeout 200 python bench.py --global-batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_4pairs.json"
timeout 200 python bench.py --config kitti_n5000_b16 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_kitti_n5000_b16.json"
timeout 200 python bench.py --config lomatch_n10000_b8 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_lomatch_n10000_b8.json"
timeout 200 python bench.py --config n1000_b1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1.json"
timeout 200 python bench.py --config n1000_b1 --in-flight 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n1000_b1_inflight3.json"
ls -la "$OUT" | grep r03_d
