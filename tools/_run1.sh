cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "layer or pack or split or encoder or forward" 2>&1 | tail -5 > gpurun_out/r1y_pytest_layer.txt
timeout 120 python tools/layer_trace.py --bs 4 2>&1 | sed -n 2,4p > gpurun_out/r1y_trace.txt
timeout 120 python tools/layer_trace.py --bs 32 2>&1 | sed -n 2,4p >> gpurun_out/r1y_trace.txt
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r1y_bench.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 2 --global-batch 4 > gpurun_out/r1y_bench_b4.txt 2>&1
