set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "in_flight or ragged or eight_rank or forward_is_bitwise" 2>&1 | tail -15 > $OUT/r03_f_pytest.txt
timeout 300 python tools/inflight_ab.py --batch 32 --steps 30 > $OUT/r03_f_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 4 --steps 150 >> $OUT/r03_f_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config kitti_n5000_b16 --steps 50 >> $OUT/r03_f_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config lomatch_n10000_b8 --steps 30 >> $OUT/r03_f_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --config n1000_b1 --steps 500 >> $OUT/r03_f_inflight_ab.txt 2>&1
cat $OUT/r03_f_pytest.txt | tail -5; grep -v amdgpu $OUT/r03_f_inflight_ab.txt
