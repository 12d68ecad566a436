#!/bin/bash
# Run on the GPU box (through gpurun): issue / stall / cache counters of the fused layer kernels only.
#   bash tools/gpu_pmc_layer.sh <tag> [ENV=value ...]      e.g.  bash tools/gpu_pmc_layer.sh r02j_h3 PDSC_LAYER_GEMM=1
# Separate --pmc passes with --kernel-trace only (no sys/hip/hsa tracing), summaries to gpurun_out/<tag>_summary.txt.
set -u
TAG=${1:-pmcl}
shift || true
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SUMMARY=$OUT/${TAG}_summary.txt
echo "# env: $*" > "$SUMMARY"
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" ; do
  # (TCP_* / TA_* / TD_* passes hang rocprofv3 on this image until the timeout -- 5 GPU-minutes each; left out)
  i=$((i+1))
  rm -rf /tmp/pmcl_$i
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmcl_$i -o p -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-check --sustain-seconds 0 > "$OUT/${TAG}_pass$i.log" 2>&1
  echo "## pass $i: $pass (exit $?)" >> "$SUMMARY"
  DB=$(find /tmp/pmcl_$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$ROOT/tools/rocpd_pmc_stats.py" "$DB" layer_ >> "$SUMMARY" 2>&1; else tail -n 5 "$OUT/${TAG}_pass$i.log" >> "$SUMMARY"; fi
  rm -rf /tmp/pmcl_$i
done
echo "summary -> $SUMMARY"
