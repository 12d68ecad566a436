#!/bin/bash
# builds tools/pk_f32_repro.bin (gfx950): the packed-fp32 miscount reproducer, two translation units with different SLP settings
set -e
cd "$(dirname "$0")"
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
/opt/rocm/bin/hipcc $F -c pk_f32_repro.hip -o /tmp/pk_f32_repro.o
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c pk_f32_repro_noslp.hip -o /tmp/pk_f32_repro_noslp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/pk_f32_repro.o /tmp/pk_f32_repro_noslp.o -ldl -o pk_f32_repro.bin
echo built tools/pk_f32_repro.bin
