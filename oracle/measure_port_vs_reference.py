#!/usr/bin/env python3
"""How fast is the CPU `port` bench.py times on the GPU box (the oracle in timing mode: the reference's own ATen ops) against
the unmodified reference it stands in for?  Run in the BUILD container (imports /root/reference):

    python oracle/measure_port_vs_reference.py [--threads 8] [--pairs 3]   ->  profiles/cpu_baseline_ratio.json

bench.py quotes the ratio in `cpu_baseline.sample` when it has to fall back to the port (kind "port": /root/reference does not
exist on the GPU box)."""
import argparse
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--pairs", type=int, default=3)
ap.add_argument("--config", default="n5000_b32")
a = ap.parse_args()
out = {}
for kind, extra in (("reference", []), ("port", ["--force-port"])):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker", "--config", a.config, "--cpu-pairs", str(a.pairs),
                        "--cpu-threads", str(a.threads), "--check-pairs", "0"] + extra, capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["kind"] == kind, j["kind"]
    out[kind] = round(j["pairs_per_s"], 4)
out.update(config=a.config, threads=a.threads, pairs=a.pairs, port_over_reference=round(out["port"] / out["reference"], 3),
           where="build container (8 vCPU), torch CPU")
(ROOT / "profiles" / "cpu_baseline_ratio.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out))
