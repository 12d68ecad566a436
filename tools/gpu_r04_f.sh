#!/bin/bash
# r04 GPU call F: which attention launch triggers the packed-fp32 miscount -- the library of call C (element-wise P split) or the shipped one (pairwise split)
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
echo "=== att hog = pointdsc_amd/libpointdsc_hip_r04c.so (commit def88a7: every TU without SLP, element-wise P split in the attention loop)" > $O/pk_f32_repro_old_vs_new.txt
timeout 200 tools/pk_f32_repro.bin 5000 pointdsc_amd/libpointdsc_hip_r04c.so att >> $O/pk_f32_repro_old_vs_new.txt 2>&1
echo "=== att hog = pointdsc_amd/libpointdsc_hip.so (shipped: pairwise P split, hand-packed unorm16 scale)" >> $O/pk_f32_repro_old_vs_new.txt
timeout 200 tools/pk_f32_repro.bin 5000 pointdsc_amd/libpointdsc_hip.so att >> $O/pk_f32_repro_old_vs_new.txt 2>&1
echo "=== att hog = pointdsc_amd/libpointdsc_hip_slp.so (shipped sources, SLP vectorisation on)" >> $O/pk_f32_repro_old_vs_new.txt
timeout 200 tools/pk_f32_repro.bin 5000 pointdsc_amd/libpointdsc_hip_slp.so att >> $O/pk_f32_repro_old_vs_new.txt 2>&1
echo "=== synthetic hogs" >> $O/pk_f32_repro_old_vs_new.txt
timeout 300 tools/pk_f32_repro.bin 5000 pointdsc_amd/libpointdsc_hip.so cvt_s,mix,mfma >> $O/pk_f32_repro_old_vs_new.txt 2>&1
grep -v "^problem" $O/pk_f32_repro_old_vs_new.txt
timeout 300 python -m pytest tests/test_sharding_gloo.py -m gpu -q -k "census_pair or rccl" 2>&1 | tail -3
