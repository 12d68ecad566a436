set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "four_wavefront or point_fragment or merges_up_to or layer_gemm or ragged or in_flight or golden or h3" 2>&1 | tail -15 > $OUT/r03_j_pytest.txt
cat $OUT/r03_j_pytest.txt
: > $OUT/r03_j_ab_coop.txt
timeout 200 python tools/ab_forward.py --config n1000_b1 --variants u16+PDSC_LAYER_H3_COOP=0 u16+PDSC_LAYER_H3_COOP=256 --rounds 7 --steps 200 >> $OUT/r03_j_ab_coop.txt 2>&1
timeout 200 python tools/ab_forward.py --config n5000_b32 --batch 1 --variants u16+PDSC_LAYER_H3_COOP=0 u16+PDSC_LAYER_H3_COOP=256 --rounds 7 --steps 100 >> $OUT/r03_j_ab_coop.txt 2>&1
timeout 200 python tools/ab_forward.py --config lomatch_n10000_b8 --batch 1 --variants u16+PDSC_LAYER_H3_COOP=0 u16+PDSC_LAYER_H3_COOP=512 --rounds 5 --steps 40 >> $OUT/r03_j_ab_coop.txt 2>&1
timeout 200 python tools/ab_forward.py --config n5000_b32 --batch 2 --variants u16+PDSC_LAYER_H3_COOP=0 u16+PDSC_LAYER_H3_COOP=512 --rounds 5 --steps 60 >> $OUT/r03_j_ab_coop.txt 2>&1
timeout 200 python tools/ab_forward.py --config n5000_b32 --batch 4 --variants u16+PDSC_LAYER_H3_COOP=0 u16+PDSC_LAYER_H3_COOP=1024 --rounds 5 --steps 40 >> $OUT/r03_j_ab_coop.txt 2>&1
grep -v amdgpu $OUT/r03_j_ab_coop.txt
c=n1000_b1
rm -rf /tmp/prof_$c
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o k -- python "$ROOT/bench.py" --config $c --in-flight 1 --steps 6 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --sustain-seconds 0 > /dev/null 2>&1
f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-150 > $OUT/r03_j_kernel_stats_n1000.txt; cat $OUT/r03_j_kernel_stats_n1000.txt
