#!/usr/bin/env python3
"""What instructions does the built library ship?  Extracts the gfx950 code objects from libpointdsc_hip.so (uncompressed clang
offload bundles in .hip_fatbin), disassembles them with llvm-objdump and reports the packed fp32 VALU instructions.

    python tools/isa_audit.py [path/to/lib.so]

The rule it checks (tests/test_cpu_oracle_and_abi.py::test_library_ships_no_packed_fp32_with_operand_selects): NO v_pk_*_f32
instruction with a non-default op_sel / op_sel_hi.  That form -- a scalar or a constant broadcast into both halves of a packed
operand, what the SLP vectoriser and `float2{x, x} * v` produce -- returned wrong lanes whenever the wave shared a CU with the
split attention kernel (tools/pk_f32_repro.hip, profiles/r04_pk_f32_repro*.txt); packed fp32 with default selects never did.
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib: Path):
    data = lib.read_bytes()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        (n,) = struct.unpack_from("<Q", data, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[i + off: i + off + size])
        pos = i + len(MAGIC)
    return out


def audit(lib: Path):
    """{'code_objects', 'kernels', 'pk_f32', 'pk_f32_with_selects': [(kernel, instruction), ...]}"""
    rep = {"code_objects": 0, "kernels": 0, "pk_f32": 0, "pk_f32_with_selects": []}
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(lib)):
            f = Path(td) / f"co{k}.elf"
            f.write_bytes(co)
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, check=True).stdout
            rep["code_objects"] += 1
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel = m.group(1)
                    rep["kernels"] += 1
                    continue
                m = re.search(r"\b(v_pk_[a-z0-9]+_f32)\b(.*?)(//|$)", line)
                if m:
                    rep["pk_f32"] += 1
                    if "op_sel" in m.group(2):
                        rep["pk_f32_with_selects"].append((kernel, (m.group(1) + m.group(2)).strip()))
    return rep


if __name__ == "__main__":
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "pointdsc_amd" / "libpointdsc_hip.so"
    r = audit(lib)
    print(f"{lib}: {r['code_objects']} code objects, {r['kernels']} symbols, {r['pk_f32']} v_pk_*_f32 instructions, "
          f"{len(r['pk_f32_with_selects'])} of them with op_sel / op_sel_hi")
    for kname, ins in r["pk_f32_with_selects"][:40]:
        print("   ", kname, ":", ins)
    sys.exit(1 if r["pk_f32_with_selects"] else 0)
