"""Builds pointdsc_amd/libpointdsc_hip.so (gfx950) in-tree with hipcc.

``python -m pointdsc_amd.build`` or ``__graft_entry__.build()``.  The library is pure HIP runtime
(no torch, no pybind): ``hipcc --offload-arch=gfx950 -shared -fPIC`` over ``csrc/*.hip``.
hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the tree.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libpointdsc_hip.so"
OBJ_DIR = PKG / "csrc" / "_obj"
STAMP = OBJ_DIR / "sources.sha256"

ARCH = "gfx950"
# -ffp-contract=off: the kernels spell out every fused multiply-add (fmaf) themselves, so the arithmetic
# that decides thresholds is exactly what the source says (see DESIGN.md "numerics").
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"] + os.environ.get("PDSC_HIPCC_EXTRA", "").split()     # e.g. -DPDSC_LAYER_DIAG (diagnostic kernels)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "pointdsc_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    digest = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == digest:
        return LIB
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: Path) -> Path:
        obj = OBJ_DIR / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[pointdsc_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print("[pointdsc_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    STAMP.write_text(digest)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
