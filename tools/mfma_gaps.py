#!/usr/bin/env python3
"""Static look at how a kernel's instruction stream interleaves matrix and other work: for one kernel of a `hipcc -S`
listing, the number of non-MFMA instructions between consecutive MFMAs (in-order issue: back-to-back dependent MFMAs
stall the wave for the whole pass time; ~7 vector instructions fit in the shadow of one 32x32x16 MFMA).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only csrc/layer_h3.hip -o k.s
    python tools/mfma_gaps.py k.s layer_h3_kernelILb1ELb1ELb0
"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
gaps, cur, kinds = [], 0, Counter()
n_mfma = 0
for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith("s_endpgm"):
        break
    if not s or s.startswith(";") or s.startswith(".") or s.split(";")[0].strip().endswith(":"):
        continue
    op = s.split()[0]
    if op.startswith("v_mfma"):
        if n_mfma:
            gaps.append(cur)
        n_mfma += 1
        cur = 0
    else:
        cur += 1
        kinds[re.sub(r"_e(32|64)$", "", op).split("_")[0] + "_" + (op.split("_")[1] if "_" in op else "")] += 1
print(f"{n_mfma} MFMAs; non-MFMA instructions between consecutive MFMAs:")
hist = Counter(min(g, 40) for g in gaps)
for k in sorted(hist):
    print(f"  {k if k < 40 else '40+':>3}: {hist[k]}")
print("sequence:", " ".join(str(g) for g in gaps))
