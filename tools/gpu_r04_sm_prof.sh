#!/bin/bash
# rocprofv3 kernel summary of the spectral-matching baseline's two forms (tools/sm_bench.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/r04z
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_sm
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sm -o k -- python $ROOT/tools/sm_bench.py > $ROOT/gpurun_out/r04z/rocprof_sm.log 2>&1
DB=$(find /tmp/prof_sm -name '*.db' | head -1)
[ -n "$DB" ] && python $ROOT/tools/rocpd_kernel_stats.py "$DB" > $ROOT/gpurun_out/r04z/kernel_stats_sm_bench.txt 2>&1
head -14 $ROOT/gpurun_out/r04z/kernel_stats_sm_bench.txt | cut -c1-150
