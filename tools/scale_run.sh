#!/bin/bash
# The multi-GPU scaling runs, exactly as the driver launches them (one process per GPU, RCCL over xGMI; README / DESIGN section 7).
# On an 8-GPU MI355X node:      bash tools/scale_run.sh                 -> gpurun_out/scale_<config>_n<N>.json
# Rehearsal on a 1-GPU box:     BACKEND=gloo bash tools/scale_run.sh    (all ranks share the GPU: code path only, no scaling claim)
# The data path has no collective: each rank runs the whole hot path on its contiguous shard of the batch; the one
# all_gather of the [B/ws, 16] poses rides behind each forward on the forward's stream (bench.py: InFlight(post=gather)).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
BACKEND=${BACKEND:-nccl}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in n5000_b32 kitti_n5000_b16 lomatch_n10000_b8; do
  for N in 1 2 4 8; do
    if [ "$N" = 1 ]; then
      python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --config $cfg --no-cpu-baseline 2> "$OUT/scale_${cfg}_n$N.err" | tail -1 > "$OUT/scale_${cfg}_n$N.json"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py --gpus $N --steps $STEPS --warmup $WARMUP --config $cfg --backend $BACKEND --no-cpu-baseline 2> "$OUT/scale_${cfg}_n$N.err" | tail -1 > "$OUT/scale_${cfg}_n$N.json"
    fi
    python - "$OUT/scale_${cfg}_n$N.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(f"{d['config']['name']:20s} gpus {d['n_gpus']}  pairs/GPU {d['config']['pairs_per_gpu']:3d}  {d['value']:9.1f} pairs/s  {d['ms_per_step']:8.3f} ms/step  check {d.get('check', {}).get('ok')}")
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
  done
done
