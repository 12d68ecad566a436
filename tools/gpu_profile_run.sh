#!/bin/bash
# Run on the GPU box (through gpurun): the round's evidence bundle, SHIPPED DEFAULTS -> gpurun_out/<tag>_*; copy what should be
# judged into profiles/.   bash tools/gpu_profile_run.sh <tag> [quick]
#   <tag>_pytest_gpu.txt                    tail of pytest -m gpu
#   <tag>_bench_line_<config>.json          one bench.py line per BASELINE.json configuration (check + cpu_baseline included)
#   <tag>_bench_line_<config>_<B>pairs.json the per-GPU shares of the 2- / 4- / 8-GPU runs of the sharded configurations
#   <tag>_kernel_stats_<config>[_<B>pairs].txt  rocprofv3 --kernel-trace summary of a short single-stream bench run
#   <tag>_pmc_summary.txt                   PMC passes over the headline configuration (tools/gpu_pmc_run.sh)
#   <tag>_parity_census.txt                 tools/parity_census.py: every census pair at every batch size
#   <tag>_stage_census.txt                  seeds / neighbour sets / votes of the trained-like families against the reference's recorded decisions
#   <tag>_match_bench.txt, <tag>_sm_bench.txt, <tag>_compat_bench.txt   micro-benches
set -u
TAG=${1:-prof}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
CONFIGS="n5000_b32 n1000_b1 kitti_n5000_b16 lomatch_n10000_b8"
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > "$OUT/${TAG}_pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -5 > "$OUT/${TAG}_smoke.txt"
# r06: the stage census on the trained-like checkpoints (default arithmetic and the exact-fp32 floor), one line per family
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k stage_decisions 2>&1 | grep "STAGE-CENSUS" | sed 's/^[F.]*//' > "$OUT/${TAG}_stage_census.txt"
if [ -z "$QUICK" ]; then
  # PMC first: the bench lines below quote the HBM traffic of THIS build (profiles/traffic.json is regenerated from the summary)
  bash "$ROOT/tools/gpu_pmc_run.sh" ${TAG}_pmc --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
  python tools/traffic_from_pmc.py "$OUT/${TAG}_pmc_summary.txt" n5000_b32 32 > "$OUT/${TAG}_traffic.txt" 2>&1
  cp profiles/traffic.json "$OUT/${TAG}_traffic.json"
  cd "$ROOT"
fi
for c in $CONFIGS kitti_n12000_b4 multiway_n20000_b1 trained_n5000_b32 trained_n1000_b1 trained_kitti_n5000_b16 trained_lomatch_n10000_b8; do      # (r04: the reference's real evaluation sizes; r05: trained-like weights)
  [ -f "$ROOT/tests/golden/bench_$c.npz" ] || continue
  timeout 400 python bench.py --config $c > "$OUT/${TAG}_bench_$c.log" 2>&1; tail -1 "$OUT/${TAG}_bench_$c.log" > "$OUT/${TAG}_bench_line_$c.json"
done
# r05: the same headline configuration with the per-launch key split (the fastest form: a pair's bits then depend on its batch), and the
# number an unchanged caller sees: one pair per call, result read back (bench.py --latency)
timeout 300 python bench.py --config n5000_b32 --att-leaves per_launch --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_n5000_b32_per_launch_leaves.json"
for s in n5000_b32:1 n1000_b1:1 trained_n1000_b1:1 lomatch_n10000_b8:1; do
  c=${s%%:*}
  timeout 300 python bench.py --config $c --global-batch 1 --latency --no-cpu-baseline --steps 200 --warmup 20 --sustain-seconds 1 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_${c}_latency_1pair.json"
done
SHARES="n5000_b32:16 n5000_b32:8 n5000_b32:4 kitti_n5000_b16:8 kitti_n5000_b16:4 kitti_n5000_b16:2 lomatch_n10000_b8:4 lomatch_n10000_b8:2 lomatch_n10000_b8:1"
if [ -z "$QUICK" ]; then
  for s in $SHARES; do
    c=${s%%:*}; B=${s##*:}
    timeout 300 python bench.py --config $c --global-batch $B --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_${c}_${B}pairs.json"
  done
  timeout 1200 python tools/parity_census.py --batches 0,1,2,4,8,16,32 > "$OUT/${TAG}_parity_census.txt" 2>&1
  timeout 600 python tools/parity_census.py --batches 0 --att-leaves per_launch > "$OUT/${TAG}_parity_census_per_launch_leaves.txt" 2>&1
  timeout 900 python tools/parity_census.py --families trained_n1000_b1,trained_n5000_b32,trained_kitti_n5000_b16,trained_lomatch_n10000_b8,trained_kitti_n12000_b4,trained_multiway_n20000_b1,kitti_n5000_b16,kitti_n12000_b4 --batches 0,1,2 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > "$OUT/${TAG}_parity_census_exact_fp32.txt" 2>&1
  timeout 300 python tools/attention_power.py --seconds 5 > "$OUT/${TAG}_attention_power.txt" 2>&1
  timeout 300 python tools/forward_power.py --seconds 4 > "$OUT/${TAG}_forward_power.txt" 2>&1
  timeout 200 python tools/forward_power.py --seconds 3 --pairs 4 > "$OUT/${TAG}_forward_power_4pairs.txt" 2>&1
fi
cd /tmp
for s in n5000_b32:32 n1000_b1:1 kitti_n5000_b16:16 lomatch_n10000_b8:8 n5000_b32:4 kitti_n5000_b16:2 lomatch_n10000_b8:1; do
  c=${s%%:*}; B=${s##*:}
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o k -- python "$ROOT/bench.py" --config $c --global-batch $B --in-flight 1 --steps 6 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --sustain-seconds 0 --extra off > "$OUT/${TAG}_rocprof_${c}_$B.log" 2>&1
  DB=$(find /tmp/prof_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_kernel_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_${c}_${B}pairs.txt" 2>&1
  rm -rf /tmp/prof_$c
done
if [ -z "$QUICK" ]; then
  cd "$ROOT"
  timeout 200 python tools/compat_bench.py > "$OUT/${TAG}_compat_bench.txt" 2>&1
  timeout 200 python tools/match_bench.py > "$OUT/${TAG}_match_bench.txt" 2>&1
  bash tools/gpu_run.sh ${TAG}_x knn_bench > /dev/null 2>&1; cp "$OUT/${TAG}_x/knn_bench.txt" "$OUT/${TAG}_knn_bench.txt" 2>/dev/null
  timeout 200 python tools/sm_bench.py > "$OUT/${TAG}_sm_bench.txt" 2>&1
fi
ls -la "$OUT" | grep "${TAG}_" | tail -60
