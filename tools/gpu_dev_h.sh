set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; cd $ROOT
timeout 300 python tools/coresidency_probe.py > $OUT/r03_h_coresidency_probe.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 32 --steps 30 > $OUT/r03_h_inflight_ab.txt 2>&1
timeout 300 python tools/inflight_ab.py --batch 4 --steps 150 >> $OUT/r03_h_inflight_ab.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "attention or in_flight or forward_is_bitwise or golden" 2>&1 | tail -6 > $OUT/r03_h_pytest.txt
grep -v amdgpu $OUT/r03_h_coresidency_probe.txt; grep -v amdgpu $OUT/r03_h_inflight_ab.txt; tail -3 $OUT/r03_h_pytest.txt
