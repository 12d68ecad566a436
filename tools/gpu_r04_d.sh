#!/bin/bash
# r04 GPU call D: GPU tests, census (KITTI / N=10000) vs the reference's recorded decisions, attention change A/B, KITTI stage diff
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -30 $O/pytest_gpu.txt
timeout 600 python tools/parity_census.py --only kitti_n5000_b16 --batches 16,2 > $O/census_kitti.txt 2>&1; echo "census kitti rc=$?"
timeout 600 python tools/parity_census.py --only lomatch_n10000_b8 --batches 8,1 > $O/census_lomatch.txt 2>&1; echo "census lomatch rc=$?"
timeout 600 python tools/parity_census.py --only kitti_n5000_b16 --batches 16,2 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > $O/census_kitti_exact_fp32.txt 2>&1; echo "census kitti fp32 rc=$?"
timeout 200 python tools/stage_diff.py --config kitti_n5000_b16 --pair 60 --bs 2 > $O/stage_diff_kitti_pair60.txt 2>&1
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline > $O/ab_new_$i.json 2>$O/ab_new_$i.err; echo "bench new $i rc=$?"
  POINTDSC_HIP_LIB=pointdsc_amd/libpointdsc_hip_slp.so timeout 200 python bench.py --no-cpu-baseline > $O/ab_slp_$i.json 2>$O/ab_slp_$i.err; echo "bench slp $i rc=$?"
done
grep "unexcused\|batches of" $O/census_kitti.txt $O/census_lomatch.txt $O/census_kitti_exact_fp32.txt | cut -c1-330
cat $O/stage_diff_kitti_pair60.txt | tail -9
for f in $O/ab_*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], l["value"], l["sustained"]["value"], l["single_stream"]["value"], l["roofline"]["avg_launch_ms"], l["roofline_layer"]["avg_launch_ms"], l["roofline_compat"]["avg_launch_ms"], l["check"]["ok"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
