#!/bin/bash
# Run on the GPU box (through gpurun): the round's evidence bundle -> gpurun_out/<tag>_*; copy what should be judged
# into profiles/.   bash tools/gpu_profile_run.sh <tag>
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 300 python bench.py > "$OUT/${TAG}_bench.log" 2>&1; tail -1 "$OUT/${TAG}_bench.log" > "$OUT/${TAG}_bench_line.json"
timeout 300 python bench.py --global-batch 4 --steps 20 > "$OUT/${TAG}_bench_b4.log" 2>&1; tail -1 "$OUT/${TAG}_bench_b4.log" > "$OUT/${TAG}_bench_line_4pairs.json"
for B in 8 16; do   # the per-GPU shares of the 4- and 2-GPU runs of the 32-pair configuration
  timeout 300 python bench.py --global-batch $B --steps 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_${B}pairs.json"
done
cd /tmp
for B in 32 4; do
  rm -rf /tmp/prof_$B
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$B -o k -- python "$ROOT/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --global-batch $B > "$OUT/${TAG}_rocprof_B$B.log" 2>&1
  DB=$(find /tmp/prof_$B -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_kernel_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_bench_N5000_B$B.txt" 2>&1
  rm -rf /tmp/prof_$B
done
bash "$ROOT/tools/gpu_pmc_run.sh" ${TAG}_pmc > /dev/null 2>&1
ls -la "$OUT" | tail -20
