#!/bin/bash
# Run on the GPU box (through gpurun): the round's evidence bundle -> gpurun_out/<tag>_*; copy what should be judged
# into profiles/.   bash tools/gpu_profile_run.sh <tag> [quick]
#   <tag>_pytest_gpu.txt                 tail of pytest -m gpu
#   <tag>_bench_line_<config>.json       one bench.py line per BASELINE.json configuration (check + cpu_baseline included)
#   <tag>_kernel_stats_<config>.txt      rocprofv3 --kernel-trace summary of a short bench run of that configuration
#   <tag>_pmc_summary.txt                PMC passes over the headline configuration (tools/gpu_pmc_run.sh)
#   <tag>_match_bench.txt, <tag>_sm_bench.txt   f-2 / f-3 micro-benches
set -u
TAG=${1:-prof}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
CONFIGS="n5000_b32 n1000_b1 kitti_n5000_b16 lomatch_n10000_b8"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$OUT/${TAG}_pytest_gpu.txt"
for c in $CONFIGS; do
  timeout 400 python bench.py --config $c > "$OUT/${TAG}_bench_$c.log" 2>&1; tail -1 "$OUT/${TAG}_bench_$c.log" > "$OUT/${TAG}_bench_line_$c.json"
done
if [ -z "$QUICK" ]; then
  for B in 4 8 16; do   # the per-GPU shares of the 8-, 4- and 2-GPU runs of the 32-pair configuration
    timeout 300 python bench.py --global-batch $B --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_${B}pairs.json"
  done
fi
cd /tmp
for c in $CONFIGS; do
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o k -- python "$ROOT/bench.py" --config $c --steps 6 --warmup 1 --no-cpu-baseline --no-check --sustain-seconds 0 > "$OUT/${TAG}_rocprof_$c.log" 2>&1
  DB=$(find /tmp/prof_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_kernel_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_$c.txt" 2>&1
  rm -rf /tmp/prof_$c
done
if [ -z "$QUICK" ]; then
  bash "$ROOT/tools/gpu_pmc_run.sh" ${TAG}_pmc > /dev/null 2>&1
  cd "$ROOT"
  timeout 200 python tools/match_bench.py > "$OUT/${TAG}_match_bench.txt" 2>&1
  timeout 200 python tools/sm_bench.py > "$OUT/${TAG}_sm_bench.txt" 2>&1
fi
ls -la "$OUT" | tail -30
