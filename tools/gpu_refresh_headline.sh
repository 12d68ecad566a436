#!/bin/bash
# After a late kernel change: the part of tools/gpu_profile_run.sh whose figures depend on the library build -- GPU tests, the PMC passes
# (profiles/traffic.json is stamped with the library digest), the headline bench line, the latency lines, two kernel summaries.
set -u
TAG=r06
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
timeout 1000 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > "$OUT/${TAG}_pytest_gpu.txt"; tail -2 "$OUT/${TAG}_pytest_gpu.txt"
bash "$ROOT/tools/gpu_pmc_run.sh" ${TAG}_pmc --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
cd "$ROOT"
python tools/traffic_from_pmc.py "$OUT/${TAG}_pmc_summary.txt" n5000_b32 32 > "$OUT/${TAG}_traffic.txt" 2>&1
cp profiles/traffic.json "$OUT/${TAG}_traffic.json"
timeout 400 python bench.py --config n5000_b32 > "$OUT/${TAG}_bench_n5000_b32.log" 2>&1; tail -1 "$OUT/${TAG}_bench_n5000_b32.log" > "$OUT/${TAG}_bench_line_n5000_b32.json"
for s in n5000_b32:1 n1000_b1:1 trained_n1000_b1:1 lomatch_n10000_b8:1; do
  c=${s%%:*}
  timeout 300 python bench.py --config $c --global-batch 1 --latency --no-cpu-baseline --steps 200 --warmup 20 --sustain-seconds 1 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line_${c}_latency_1pair.json"
done
cd /tmp
for s in n1000_b1:1 n5000_b32:32; do
  c=${s%%:*}; B=${s##*:}
  rm -rf /tmp/prof_$c
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o k -- python "$ROOT/bench.py" --config $c --global-batch $B --in-flight 1 --steps 6 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --sustain-seconds 0 --extra off > "$OUT/${TAG}_rocprof_${c}_$B.log" 2>&1
  DB=$(find /tmp/prof_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_kernel_stats.py" "$DB" > "$OUT/${TAG}_kernel_stats_${c}_${B}pairs.txt" 2>&1
  rm -rf /tmp/prof_$c
done
cd "$ROOT"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_bench_line_*latency*.json"))+["gpurun_out/r06_bench_line_n5000_b32.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["check"]["ok"], d["roofline_layer"].get("traffic"))
PY
grep "fixup_kabsch\|select_refine" gpurun_out/r06_kernel_stats_n1000_b1_1pairs.txt | cut -c1-120
