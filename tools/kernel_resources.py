#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel the built library ships, read from the code objects' metadata notes.

    python tools/kernel_resources.py [path/to/lib.so] [substring ...]

(.vgpr_count / .agpr_count / .sgpr_count / .private_segment_fixed_size = scratch bytes per lane / .group_segment_fixed_size = static
LDS.)  What DESIGN.md's register and spill figures are read from."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
from isa_audit import code_objects  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def resources(lib: Path):
    rows = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = g("name")
            try:
                name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
            except Exception:       # noqa: BLE001
                pass
            rows.append({"name": name, "vgpr": g("vgpr_count"), "agpr": blk.split()[0], "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size"),
                         "lds": g("group_segment_fixed_size"), "spill_vgpr": g("vgpr_spill_count"), "spill_sgpr": g("sgpr_spill_count")})
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = Path(args.pop(0)) if args and args[0].endswith(".so") else ROOT / "pointdsc_amd" / "libpointdsc_hip.so"
    for r in resources(lib):
        if args and not any(a in r["name"] for a in args):
            continue
        print(f"vgpr {r['vgpr']:>4} agpr {r['agpr']:>4} sgpr {r['sgpr']:>4} scratch {r['scratch']:>5} B spills v{r['spill_vgpr']}/s{r['spill_sgpr']} lds {r['lds']:>6}  {r['name'][:150]}")
