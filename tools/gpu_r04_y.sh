#!/bin/bash
# r04 y: GPU_MAX_HW_QUEUES (HIP maps streams onto 4 hardware queues by default; 2 forwards in flight + 2 tail streams + the current stream = 5)
mkdir -p gpurun_out/r04y
cd /root/repo
export TMPDIR=/tmp
run() { # tag config batch
  for k in 1 2; do
    for q in default 8; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      python bench.py --config $2 --global-batch $3 --no-cpu-baseline --sustain-seconds 1 > gpurun_out/r04y/$1_q${q}_$k.json 2>/dev/null
    done
  done
}
run n5000x32 n5000_b32 32
run n5000x4 n5000_b32 4
run kittix2 kitti_n5000_b16 2
run lomatchx1 lomatch_n10000_b8 1
unset GPU_MAX_HW_QUEUES
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04y/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1].ljust(28), round(d["value"],1), d["ms_per_step"], round(d["sustained"]["value"],1), round(d["single_stream"]["value"],1), d["check"]["ok"], (d.get("power") or {}).get("mean_w"))
    except Exception as e: print(f, "ERR", e)
PY
