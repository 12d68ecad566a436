#!/bin/bash
# r04 GPU call A: packed-fp32 reproducer, baseline bench of this box, census (n1000) against the reference's recorded decisions, GPU tests
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout 400 tools/pk_f32_repro.bin 400 pointdsc_amd/libpointdsc_hip.so > $O/pk_f32_repro.txt 2>&1; echo "repro rc=$?"
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python tools/parity_census.py --only n1000_b1 --batches 1,16 > $O/census_n1000.txt 2>&1; echo "census rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_gpu.txt; cat $O/pk_f32_repro.txt; cat $O/bench_default.json | head -c 1500
