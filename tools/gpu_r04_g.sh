#!/bin/bash
# r04 GPU call G: the synthetic neighbour that triggers the packed-fp32 miscount, one ingredient removed at a time
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 400 tools/pk_f32_repro.bin 3000 pointdsc_amd/libpointdsc_hip.so mix,mix-mfma,mix-exp,mix-cvt,mix_vcvt,mix-lds > $O/pk_f32_repro_mix.txt 2>&1
cat $O/pk_f32_repro_mix.txt
