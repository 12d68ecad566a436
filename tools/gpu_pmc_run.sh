#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 PMC passes over a short bench run, summarised to text on the
# box (the rocpd .db files are tens of MB each and are deleted; gpurun_out/ is capped at 64 MiB).
#   bash tools/gpu_pmc_run.sh [tag] [extra bench args...]
# Counters are collected in separate passes with --kernel-trace only (no sys/hip/hsa tracing), as required.
set -u
TAG=${1:-pmc}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SUMMARY=$OUT/${TAG}_summary.txt
: > "$SUMMARY"
FILTER="sc_attention compat layer_ knn_select knn_fused normalize_conf seed_solve nms_ gram_rows"
i=0
for pass in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmc_$i -o p -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-check --sustain-seconds 0 --extra off "$@" > "$OUT/${TAG}_pass$i.log" 2>&1
  echo "## pass $i: $pass (exit $?)" >> "$SUMMARY"
  DB=$(find /tmp/pmc_$i -name '*.db' | head -1)
  if [ -n "$DB" ]; then python "$ROOT/tools/rocpd_pmc_stats.py" "$DB" $FILTER >> "$SUMMARY" 2>&1; fi
  rm -rf /tmp/pmc_$i
done
tail -n 3 "$OUT/${TAG}_pass1.log" | cut -c1-400
echo "summary -> $SUMMARY"
