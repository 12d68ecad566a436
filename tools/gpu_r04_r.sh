#!/bin/bash
mkdir -p gpurun_out/r04r
cd /root/repo
export TMPDIR=/tmp
python bench.py > gpurun_out/r04r/bench_n5000_b32.json 2> gpurun_out/r04r/bench_n5000_b32.err; echo "bench rc=$?"
python bench.py --config n1000_b1 > gpurun_out/r04r/bench_n1000_b1.json 2> gpurun_out/r04r/bench_n1000_b1.err; echo "bench n1000 rc=$?"
python bench.py --config lomatch_n10000_b8 > gpurun_out/r04r/bench_lomatch.json 2> gpurun_out/r04r/bench_lomatch.err; echo "bench lomatch rc=$?"
timeout 600 python -m pytest tests/test_sharding_gloo.py -x -q -m gpu > gpurun_out/r04r/pytest_sharding.txt 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r04r/pytest_sharding.txt
python - <<'PY'
import json
for f in ("bench_n5000_b32","bench_n1000_b1","bench_lomatch"):
    try:
        d=json.loads(open(f"gpurun_out/r04r/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("sustained",{}).get("value"), json.dumps(d.get("power")), d["check"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
