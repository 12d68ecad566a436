"""Builds pointdsc_amd/libpointdsc_hip.so (gfx950) in-tree with hipcc.

``python -m pointdsc_amd.build`` or ``__graft_entry__.build()``.  The library is pure HIP runtime
(no torch, no pybind): ``hipcc --offload-arch=gfx950 -shared -fPIC`` over ``csrc/*.hip``.
hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the tree.

``python -m pointdsc_amd.build --experiments`` builds ``libpointdsc_hip_exp.so`` from the same sources with
``-DPDSC_EXPERIMENTS``: the PDSC_* environment knobs are live and the opt-in record kernels (64-query attention,
persistent / compat-in-registers attention, all-split layer kernel, r02 exact-rounded unorm16 compat build) are compiled
in.  Tools select it with ``POINTDSC_HIP_LIB=pointdsc_amd/libpointdsc_hip_exp.so`` (tools/ab_forward.py --exp); the product
library reads no environment variable.
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
EXP_LIB = PKG / "libpointdsc_hip_exp.so"
EXP_OBJ_DIR = PKG / "csrc" / "_obj_exp"

ARCH = "gfx950"
# -ffp-contract=off: the kernels spell out every fused multiply-add (fmaf) themselves, so the arithmetic
# that decides thresholds is exactly what the source says (see DESIGN.md "numerics").
# -fno-slp-vectorize, EVERY translation unit (r04): the SLP vectoriser pairs adjacent scalar fp32 operations into v_pk_*_f32
# with op_sel / op_sel_hi operand selects, and packed fp32 WITH such a select returns wrong lanes whenever a co-resident wave
# interleaves MFMAs with vector work: 100 % of the launches beside a synthetic neighbour, 14-25 % beside the r03 attention kernel
# (tools/pk_f32_repro.hip, no PointDSC forward; profiles/r04_*pk_f32_repro*.txt) -- scalar fp32 and packed fp32 with default selects:
# never.  The mechanism below the ISA is not established, so no instruction of that form is shipped at all: no compiler pairing
# anywhere, the hand-written float2 math of compat.hip keeps its broadcasts in register pairs, and tools/isa_audit.py checks the
# built code objects in the CPU test suite.  `--slp` builds the old flags as libpointdsc_hip_slp.so for A/B runs.
NO_SLP = ["-fno-slp-vectorize"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"] + os.environ.get("PDSC_HIPCC_EXTRA", "").split()     # e.g. -DPDSC_LAYER_DIAG (diagnostic kernels)
SLP_LIB = PKG / "libpointdsc_hip_slp.so"
SLP_OBJ_DIR = PKG / "csrc" / "_obj_slp"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


# extra flags of single sources.  score_slp.hip (experiments builds): the r03 reproducer, deliberately WITH SLP vectorisation
PER_FILE_FLAGS = {"score_slp.hip": ["-fslp-vectorize"]}


def _sources(experiments: bool = False):
    """csrc/*.hip is the product library; csrc/experiments/*.hip (A/B reproducers such as the SLP-vectorised scoring kernel) joins
    experiments builds only."""
    return sorted(CSRC.glob("*.hip")) + (sorted((CSRC / "experiments").glob("*.hip")) if experiments else [])


def _digest(flags=None) -> str:
    flags = FLAGS if flags is None else flags
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list((CSRC / "experiments").glob("*.hip")) +
                    [PKG.parent / "include" / "pointdsc_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(flags).encode())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def source_digest() -> str:
    """sha256 over the product library's sources (csrc/*.hip, csrc/*.h, include/pointdsc_hip.h) -- what identifies "this build" for
    evidence files collected on the GPU box (profiles/traffic.json)."""
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "pointdsc_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, experiments: bool = False, slp: bool = False) -> Path:
    flags = FLAGS + ([] if slp else NO_SLP) + (["-DPDSC_EXPERIMENTS"] if experiments else [])
    lib, obj_dir = (SLP_LIB, SLP_OBJ_DIR) if slp else ((EXP_LIB, EXP_OBJ_DIR) if experiments else (LIB, OBJ_DIR))
    stamp = obj_dir / "sources.sha256"
    digest = _digest(flags)
    if not force and lib.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        return lib
    obj_dir.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        # per-file digest: only sources whose text (or any header) changed are recompiled
        fd = hashlib.sha256(src.read_bytes() + b"".join(p.read_bytes() for p in sorted(CSRC.glob("*.h"))) +
                            (PKG.parent / "include" / "pointdsc_hip.h").read_bytes() +
                            " ".join(flags + PER_FILE_FLAGS.get(src.name, [])).encode()).hexdigest()
        fstamp = obj_dir / (src.stem + ".sha256")
        if not force and obj.exists() and fstamp.exists() and fstamp.read_text().strip() == fd:
            return obj
        cmd = [hipcc, *flags, *PER_FILE_FLAGS.get(src.name, []), f"-I{CSRC}", "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[pointdsc_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        fstamp.write_text(fd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources(experiments)))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(lib)]
    if verbose:
        print("[pointdsc_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(digest)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv, slp="--slp" in sys.argv))
