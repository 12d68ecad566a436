set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
: > $OUT/r03_ae_inflight_ab.txt
timeout 300 python tools/inflight_ab.py --config n1000_b1 --steps 400 >> $OUT/r03_ae_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 1 --steps 200 >> $OUT/r03_ae_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 4 --steps 100 >> $OUT/r03_ae_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config kitti_n5000_b16 --batch 2 --steps 150 >> $OUT/r03_ae_inflight_ab.txt 2>&1
grep -v amdgpu $OUT/r03_ae_inflight_ab.txt
