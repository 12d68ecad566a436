#!/bin/bash
# r04 GPU call E: bench.py's check on a census pair outside the contract, bench lines of the large-N workloads, power record with the energy counter
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 200 python bench.py --config kitti_n5000_b16 --global-batch 2 --first-pair 60 --no-cpu-baseline > $O/bench_kitti_pair60.json 2>$O/bench_kitti_pair60.err; echo "rc=$?"
timeout 300 python bench.py --config kitti_n12000_b4 > $O/bench_line_kitti_n12000_b4.json 2>$O/bench_kitti_n12000_b4.err; echo "rc=$?"
timeout 300 python bench.py --config multiway_n20000_b1 > $O/bench_line_multiway_n20000_b1.json 2>$O/bench_multiway.err; echo "rc=$?"
timeout 200 python tools/attention_power.py --seconds 6 > $O/attention_power.txt 2>&1; echo "rc=$?"
python - <<'PY'
import json
for f in ("bench_kitti_pair60","bench_line_kitti_n12000_b4","bench_line_multiway_n20000_b1"):
    try:
        l=json.loads(open(f"gpurun_out/r04e/{f}.json").read().strip().splitlines()[-1])
        print(f, l["value"], l["ms_per_step"], l.get("sustained",{}).get("value"), l["roofline"]["executed_frac"], l["roofline_compat"]["frac"], json.dumps(l["check"])[:900], l.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
tail -4 gpurun_out/r04e/attention_power.txt; tail -3 gpurun_out/r04e/bench_kitti_pair60.err
