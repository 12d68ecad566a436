cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r1x_pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > gpurun_out/r1x_smoke.txt
timeout 300 python bench.py > gpurun_out/r1x_bench.txt 2>&1
timeout 300 python bench.py --global-batch 4 --steps 20 > gpurun_out/r1x_bench_b4.txt 2>&1
