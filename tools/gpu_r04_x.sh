#!/bin/bash
# r04 x: final sanity of the last code state: GPU tests, smoke, the n1000_b1 line with the new small-step defaults, the headline line
mkdir -p gpurun_out/r04x
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04x/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04x/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
python bench.py --config n1000_b1 > gpurun_out/r04x/bench_line_n1000_b1.json 2> gpurun_out/r04x/bench_n1000.err; echo "bench n1000 rc=$?"
python bench.py > gpurun_out/r04x/bench_line_n5000_b32.json 2> gpurun_out/r04x/bench_n5000.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("bench_line_n1000_b1","bench_line_n5000_b32"):
    d=json.loads(open(f"gpurun_out/r04x/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["sustained"]["value"], d["single_stream"]["value"], d["in_flight"], d["hip_graphs"], d.get("zero_copy_graphs"), d["check"]["ok"], (d.get("power") or {}).get("mean_w"), d["roofline"]["avg_launch_ms"])
PY
