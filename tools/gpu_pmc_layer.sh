#!/bin/bash
# Run on the GPU box (through gpurun): SQ / TCP / TCC counters of the fused layer kernel and its neighbours.
#   bash tools/gpu_pmc_layer.sh [tag] [extra bench args...]
set -u
TAG=${1:-pmc_layer}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SUMMARY=$OUT/${TAG}_summary.txt
: > "$SUMMARY"
FILTER="layer_ sc_attention"
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_STALL_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmcl_$i
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmcl_$i -o p -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$OUT/${TAG}_pass$i.log" 2>&1
  echo "## pass $i: $pass (exit $?)" >> "$SUMMARY"
  DB=$(find /tmp/pmcl_$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$ROOT/tools/rocpd_pmc_stats.py" "$DB" $FILTER >> "$SUMMARY" 2>&1; else tail -5 "$OUT/${TAG}_pass$i.log" >> "$SUMMARY"; fi
  rm -rf /tmp/pmcl_$i
done
echo "summary -> $SUMMARY"
