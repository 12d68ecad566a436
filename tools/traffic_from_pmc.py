#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/gpu_pmc_run.sh -> <tag>_summary.txt).

    python tools/traffic_from_pmc.py gpurun_out/r03_pmc_summary.txt n5000_b32 32 [--out profiles/traffic.json]

For every kernel of the summary: HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KiB per dispatch; on
gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes -- MI355X_MICROARCH.md, HBM section -- so
it is doubled; WRITE_SIZE is used as reported: it matches the compat kernel's known byte count to 1 %).  FETCH_SIZE and
WRITE_SIZE come from separate passes.  bench.py copies the figures of the dominant kernels into `roofline*.traffic`.
"""
import json
import re
import sys
from pathlib import Path

SHORT = {"sc_attention_split_kernel": "sc_attention_split_kernel", "compat_sym_u16_kernel": "compat_sym_u16_kernel",
         "compat_sym_kernel": "compat_sym_kernel", "layer_h3_kernel<true, true": "layer_h3_kernel", "layer_h3_coop_kernel<true, true": "layer_h3_coop_kernel",
         "layer_wave_kernel<true, true": "layer_wave_kernel", "layer_fused_kernel<true, true": "layer_fused_kernel",
         "gram_rows_kernel": "gram_rows_kernel", "knn_select_kernel": "knn_select_kernel", "knn_fused_kernel": "knn_fused_kernel",
         "seed_solve_kernel": "seed_solve_kernel"}


def main():
    summary, config, pairs = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = Path(sys.argv[sys.argv.index("--out") + 1]) if "--out" in sys.argv else Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
    kernel, vals = None, {}
    for line in Path(summary).read_text().splitlines():
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+) avg_duration_us=([\d.]+)", line)
        if m:
            kernel = next((v for k, v in SHORT.items() if k in m.group(1)), None)
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+avg=([\d.e+]+)", line)
        if m and kernel:
            vals.setdefault(kernel, {})[m.group(1)] = float(m.group(2))
    entry = {"_source": str(summary), "_detail": {}}
    for k, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            entry[k] = int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
            entry["_detail"][k] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"]}
    tj = json.loads(out.read_text()) if out.exists() else {}
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from pointdsc_amd import build as _build
    digest = _build.source_digest()
    if tj.get("_library_source_sha256") != digest:
        tj = {}          # entries collected on another build of the library are not this build's traffic: start over
    tj["_how"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --in-flight 1` (tools/gpu_pmc_run.sh, "
                  "tools/traffic_from_pmc.py); KiB per dispatch averaged over the dispatches of a kernel; traffic = 2 x FETCH_SIZE + "
                  "WRITE_SIZE bytes per launch (gfx950: FETCH_SIZE counts wide streaming reads at half their bytes, MI355X_MICROARCH.md)")
    tj[f"{config}_B{pairs}_u16"] = entry
    # which library the counters were collected on: bench.py uses the file only when this equals the digest of the sources it runs
    tj["_library_source_sha256"] = digest
    tj["_collected"] = str(summary)
    out.write_text(json.dumps(tj, indent=1))
    print(json.dumps({k: v for k, v in entry.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
