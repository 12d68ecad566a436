#!/bin/bash
# r06: energy budget of the attention launch from compile-time knock-outs of its MFMAs (PDSC_ATT_ABLATE, see
# attention_split.hip: WRONG results, timing / power only).  Build the ablation libraries first (build container):
#   for m in 1 3 7; do PDSC_HIPCC_EXTRA=-DPDSC_ATT_ABLATE=$m python -m pointdsc_amd.build; cp pointdsc_amd/libpointdsc_hip.so pointdsc_amd/libpointdsc_hip_abl$m.so; done
#   python -m pointdsc_amd.build
# then on the GPU box:  bash tools/attention_energy_budget.sh  -> gpurun_out/r06_attention_energy_budget.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_attention_energy_budget.txt
mkdir -p "$ROOT/gpurun_out"; cd "$ROOT"; : > "$OUT"
for rep in 1 2; do
for m in 0 1 3 7; do
  lib=$ROOT/pointdsc_amd/libpointdsc_hip.so; [ $m != 0 ] && lib=$ROOT/pointdsc_amd/libpointdsc_hip_abl$m.so
  [ -f "$lib" ] || continue
  echo "## ablation mask $m (MFMAs left: $( [ $m = 0 ] && echo 6/6 || ( [ $m = 1 ] && echo 5/6 || ( [ $m = 3 ] && echo 3/6 || echo 2/6 ) ) )), repetition $rep" >> "$OUT"
  POINTDSC_HIP_LIB=$lib timeout 120 python tools/attention_power.py --seconds 3 2>&1 | grep '"arm"' | grep -v idle | cut -c1-420 >> "$OUT"
done
done
cat "$OUT" | cut -c1-330
