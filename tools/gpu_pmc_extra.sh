#!/bin/bash
# PMC passes (tools/gpu_pmc_run.sh) over the other BASELINE.json configurations and the 8-GPU share of the headline one.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
TAG=${1:-r03}
bash tools/gpu_pmc_run.sh ${TAG}_pmc_kitti_n5000_b16 --config kitti_n5000_b16 --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
bash tools/gpu_pmc_run.sh ${TAG}_pmc_lomatch_n10000_b8 --config lomatch_n10000_b8 --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
bash tools/gpu_pmc_run.sh ${TAG}_pmc_n5000_4pairs --global-batch 4 --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
bash tools/gpu_pmc_run.sh ${TAG}_pmc_n1000_b1 --config n1000_b1 --in-flight 1 --settle-seconds 0 > /dev/null 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "in_flight or harness or ragged" 2>&1 | tail -8 > gpurun_out/${TAG}_pytest_rerun.txt
cat gpurun_out/${TAG}_pytest_rerun.txt; ls gpurun_out | grep "${TAG}_pmc_.*summary"
