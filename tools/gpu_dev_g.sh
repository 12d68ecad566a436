set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
timeout 300 python tools/inflight_ab.py --batch 32 --steps 30 > $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 4 --steps 150 >> $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config kitti_n5000_b16 --steps 50 >> $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config kitti_n5000_b16 --batch 2 --steps 200 >> $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config lomatch_n10000_b8 --steps 30 >> $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config lomatch_n10000_b8 --batch 1 --steps 150 >> $OUT/r03_g_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config n1000_b1 --steps 500 >> $OUT/r03_g_inflight_ab.txt 2>&1
grep -v amdgpu $OUT/r03_g_inflight_ab.txt
