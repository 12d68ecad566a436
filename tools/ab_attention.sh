#!/bin/bash
# A/B of the attention launch time (hipEvents inside bench.py) over per-GPU batch, compat format and the nt policy.
#   bash tools/ab_attention.sh <out-file>
OUT=${1:-gpurun_out/ab_attention.txt}
mkdir -p $(dirname $OUT); : > $OUT
for B in 2 4 8 32; do for F in f32 u16; do for NT in 0 1; do
  POINTDSC_COMPAT_FORMAT=$F PDSC_ATT_COMPAT_NT=$NT timeout 300 python bench.py --global-batch $B --steps 20 --no-cpu-baseline --no-check --sustain-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=$B fmt=$F nt=$NT pairs/s=%.1f att_ms=%.4f per_pair_us=%.2f layer_ms=%.4f compat_ms=%.4f' % (d['value'], r['avg_launch_ms'], r['avg_launch_ms']*1e3/$B, d['roofline_layer']['avg_launch_ms'], d['roofline_compat']['avg_launch_ms']))" >> $OUT
done; done; done
cat $OUT
