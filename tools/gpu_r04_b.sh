#!/bin/bash
# r04 GPU call B: reproducer statistics, SLP on/off A/B, census vs the reference's recorded decisions (n1000, n5000), stage diffs, power record, GPU tests
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 300 tools/pk_f32_repro.bin 20000 pointdsc_amd/libpointdsc_hip.so att,att32,mfma > $O/pk_f32_repro_stats.txt 2>&1; echo "repro rc=$?"
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline > $O/ab_noslp_$i.json 2>/dev/null; echo "bench noslp $i rc=$?"
  POINTDSC_HIP_LIB=pointdsc_amd/libpointdsc_hip_slp.so timeout 200 python bench.py --no-cpu-baseline > $O/ab_slp_$i.json 2>/dev/null; echo "bench slp $i rc=$?"
done
timeout 600 python tools/parity_census.py --only n1000_b1 --batches 1,16 > $O/census_n1000.txt 2>&1; echo "census n1000 rc=$?"
timeout 600 python tools/parity_census.py --only n5000_b32 --batches 32,4,2 > $O/census_n5000.txt 2>&1; echo "census n5000 rc=$?"
timeout 600 python tools/parity_census.py --only n1000_b1 --batches 1,16 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > $O/census_n1000_exact_fp32.txt 2>&1; echo "census n1000 fp32 rc=$?"
for p in 126 150 206 69 44; do timeout 120 python tools/stage_diff.py --config n1000_b1 --pair $p > $O/stage_diff_n1000_pair$p.txt 2>&1; done; echo stage diffs done
timeout 200 python tools/attention_power.py --wide --seconds 3 > $O/attention_power.txt 2>&1; echo "power rc=$?"
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.txt; cat $O/pk_f32_repro_stats.txt; cat $O/attention_power.txt | tail -12
for f in $O/ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["sustained"]["value"], l["single_stream"]["value"], l["roofline"]["avg_launch_ms"], l["roofline_layer"]["avg_launch_ms"], l["roofline_compat"]["avg_launch_ms"], l["check"]["ok"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
