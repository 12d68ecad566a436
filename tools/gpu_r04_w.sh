#!/bin/bash
# r04 w: n1000_b1 (host / queue bound): in-flight depth x GPU_MAX_HW_QUEUES
mkdir -p gpurun_out/r04w
cd /root/repo
export TMPDIR=/tmp
for q in default 8; do
  for d in 2 3 4 5 6 8; do
    for k in 1 2; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      python bench.py --config n1000_b1 --no-cpu-baseline --in-flight $d --sustain-seconds 1 > gpurun_out/r04w/b_q${q}_d${d}_$k.json 2>/dev/null
    done
  done
done
unset GPU_MAX_HW_QUEUES
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04w/b_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(24), round(d["value"]), d["ms_per_step"], round(d["sustained"]["value"]), round(d["single_stream"]["value"]), d["in_flight"], d["check"]["ok"])
    except Exception as e: print(f, "ERR", e)
PY
