#!/bin/bash
# r04 v: zero-copy hipGraph replays (n1000_b1 is host-bound): bench with / without, the graph tests, the default bench line
mkdir -p gpurun_out/r04v
cd /root/repo
export TMPDIR=/tmp
for k in 1 2; do
  python bench.py --config n1000_b1 --no-cpu-baseline > gpurun_out/r04v/bench_n1000_zc_$k.json 2> gpurun_out/r04v/bench_n1000_zc_$k.err; echo "zc $k rc=$?"
  python bench.py --config n1000_b1 --no-cpu-baseline --zero-copy off > gpurun_out/r04v/bench_n1000_copy_$k.json 2> gpurun_out/r04v/bench_n1000_copy_$k.err; echo "copy $k rc=$?"
done
python bench.py --config n1000_b1 --no-cpu-baseline --in-flight 6 > gpurun_out/r04v/bench_n1000_zc_d6.json 2>/dev/null; echo "zc d6 rc=$?"
python bench.py --config n1000_b1 --no-cpu-baseline --in-flight 3 > gpurun_out/r04v/bench_n1000_zc_d3.json 2>/dev/null; echo "zc d3 rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hipgraph or in_flight or inflight" > gpurun_out/r04v/pytest_graphs.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04v/pytest_graphs.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04v/bench_n1000_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), d["value"], d["ms_per_step"], d["sustained"]["value"], d["single_stream"]["value"], d["in_flight"], d["hip_graphs"], d.get("zero_copy_graphs"), d["check"]["ok"], (d.get("power") or {}).get("mean_w"))
    except Exception as e: print(f, "ERR", e)
PY
