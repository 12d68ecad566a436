#!/bin/bash
# One parametrised GPU-box script (through gpurun) instead of a one-off script per experiment:
#     bash tools/gpu_run.sh <tag> <step> [<step> ...]        -> gpurun_out/<tag>/...
# steps:
#   newtests        the r05 tests only (merged attention, canonical leaves, SM opt-in, trained census), fail-fast
#   tests           pytest -m gpu, whole suite (tail of the log kept)
#   smoke           __graft_entry__.smoke()
#   ab_leaves       bench lines of n5000_b32 at 32 / 4 / 1 pairs for att_leaves = per_launch | canonical | 2 | 8
#   ab_leaves_more  the same A/B on kitti_n5000_b16 (16, 2), lomatch_n10000_b8 (8, 1), n1000_b1 (1)
#   latency         one pair per call, result read back (bench.py --latency): n5000 x1, n2000-like, n1000 x1
#   census_trained  tools/parity_census.py on the trained-like families, every batch size, default and exact-fp32 arithmetic
#   census          tools/parity_census.py on every family (default arithmetic), batches 0,1,2,4,8,16,32
#   bundle          tools/gpu_profile_run.sh <tag> (the round's evidence bundle: bench lines, rocprof, PMC, micro-benches)
#   knn_bench       kNN of the seeds: two-launch (S x N matrix) form against the fused form
#   match_bench     tools/match_bench.py (f-2 correspondence construction)
#   kitti_stage     which arithmetic moves the KITTI pairs 60 / 21 / 26: one knob at a time
#   kitti_stage_trained  the same on the trained-like KITTI family, with this library's logits against the reference's recorded ones
#   ab_lib          same-box A/B of this build against pointdsc_amd/libpointdsc_hip_prev.so (interleaved bench lines)
#   att_err         one split-precision attention launch against the fp64 softmax (sets the bounds of the stage test)
#   census_multiway tools/parity_census.py on the large-N seeded families
set -u
TAG=${1:?tag}
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
line() { # file, bench args...
  local f=$1; shift
  timeout 400 python bench.py "$@" > "$OUT/$f.log" 2>&1
  tail -1 "$OUT/$f.log" > "$OUT/$f.json"
}
summ() {
python - "$OUT" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rl, ra = d.get("roofline_layer") or {}, d.get("roofline") or {}
        print(os.path.basename(f).ljust(44), "value", round(d["value"], 1), "ms", d["ms_per_step"], "sust", round((d.get("sustained") or {}).get("value") or 0, 1),
              "single", round((d.get("single_stream") or {}).get("value") or 0, 1), "att_ms", ra.get("avg_launch_ms"), "att_frac", ra.get("frac"),
              "lay_ms", rl.get("avg_launch_ms"), "lay_frac", rl.get("frac"), "ok", (d.get("check") or {}).get("ok"), "plan", (d.get("config") or {}).get("attention_plan"))
    except Exception as e:
        print(os.path.basename(f), "ERR", repr(e)[:100])
PY
}
for STEP in "$@"; do
case $STEP in
newtests)
  timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --tb=short -k "leaves or leaf or resident or trained or knn_fused or did_not_write or large_ragged" 2>&1 | tail -40 > "$OUT/newtests.txt"
  tail -5 "$OUT/newtests.txt" ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 > "$OUT/pytest_gpu.txt"
  tail -5 "$OUT/pytest_gpu.txt" ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -5 > "$OUT/smoke.txt"; cat "$OUT/smoke.txt" ;;
ab_leaves)
  for L in per_launch canonical 2 8; do
    line ab_n5000_b32_x32_$L --config n5000_b32 --att-leaves $L --no-cpu-baseline --sustain-seconds 1.5
  done
  for L in per_launch canonical 8; do
    line ab_n5000_b32_x4_$L --config n5000_b32 --global-batch 4 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
    line ab_n5000_b32_x1_$L --config n5000_b32 --global-batch 1 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
  done
  summ | tee "$OUT/ab_leaves_summary.txt" ;;
ab_leaves_more)
  for L in per_launch canonical; do
    line ab_kitti_x16_$L --config kitti_n5000_b16 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
    line ab_kitti_x2_$L --config kitti_n5000_b16 --global-batch 2 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
    line ab_lomatch_x8_$L --config lomatch_n10000_b8 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
    line ab_lomatch_x1_$L --config lomatch_n10000_b8 --global-batch 1 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
    line ab_n1000_x1_$L --config n1000_b1 --att-leaves $L --no-cpu-baseline --sustain-seconds 1
  done
  summ | tee "$OUT/ab_leaves_more_summary.txt" ;;
latency)
  for L in per_launch canonical; do
    line lat_n5000_x1_$L --config n5000_b32 --global-batch 1 --latency --att-leaves $L --no-cpu-baseline --steps 200 --warmup 20 --sustain-seconds 1
    line lat_n1000_x1_$L --config n1000_b1 --latency --att-leaves $L --no-cpu-baseline --steps 400 --warmup 20 --sustain-seconds 1
    line lat_trained_n1000_x1_$L --config trained_n1000_b1 --latency --att-leaves $L --no-cpu-baseline --steps 400 --warmup 20 --sustain-seconds 1
  done
  summ | tee "$OUT/latency_summary.txt" ;;
ab_lib)
  # same-box A/B of two library builds: this build against another one placed at pointdsc_amd/libpointdsc_hip_prev.so (same
  # ABI version); interleaved so that board drift hits both alike.  (r05j: fp16 hi/lo operand pairs against the bf16 pairs of the
  # commit before; r05l: lo halves by v_fma_mixlo/mixhi_f16 against convert-subtract-convert)
  for R in 1 2 3; do
    line s16_this_r$R --config n5000_b32 --no-cpu-baseline --sustain-seconds 1.5
    POINTDSC_HIP_LIB=$ROOT/pointdsc_amd/libpointdsc_hip_prev.so line s16_prev_r$R --config n5000_b32 --no-cpu-baseline --sustain-seconds 1.5
  done
  line s16_this_kitti --config kitti_n5000_b16 --no-cpu-baseline --sustain-seconds 1
  POINTDSC_HIP_LIB=$ROOT/pointdsc_amd/libpointdsc_hip_prev.so line s16_prev_kitti --config kitti_n5000_b16 --no-cpu-baseline --sustain-seconds 1
  summ | tee "$OUT/ab_lib_summary.txt" ;;
census_multiway)
  timeout 900 python tools/parity_census.py --families multiway_n20000_b1,lomatch_n10000_b8,n12000_b4 2>&1 | tail -40 > "$OUT/census_multiway.txt"; cat "$OUT/census_multiway.txt" ;;
att_err)
  # measured error of the split-precision attention against the fp64 softmax (the construction of
  # tests/test_gpu_parity.py::test_sc_attention_split_matches_fp64_softmax), to set that test's bounds from
  timeout 600 python - > "$OUT/att_err.txt" 2>&1 <<'PY'
import importlib.util, sys, torch
sys.path.insert(0, "tests")
spec = importlib.util.spec_from_file_location("tgp", "tests/test_gpu_parity.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
ops, g = t.ops, t.g
for n, bs in ((257, 1), (1000, 2), (5000, 1), (1500, 9)):
    gen = torch.Generator().manual_seed(n + bs)
    batch = t.synthetic.make_batch(bs, n, seed=70 + n)
    compat = ops.spatial_compat(g(batch["src_keypts"]), g(batch["tgt_keypts"]), g(torch.tensor([0.1])))
    for qk_scale in (0.35, 2.0):
        q, k, v = (torch.randn(bs, n, 128, generator=gen) * s for s in (qk_scale, qk_scale, 1.0))
        qkv = torch.cat([q * t.QSCALE, k, v], dim=-1).reshape(bs * n, 384)
        qs, kv = ops.pack_qkv_split(g(qkv), bs, n)
        for nsplit in (1, 0, 3):
            msg = ops.sc_attention_split(qs, kv, compat, bs, n, nsplit=nsplit).cpu().reshape(bs, n, 128)
            et = em = 0.0
            for b in range(bs):
                cm = compat[b, :, :n].cpu()
                want = t._attention_ref(q[b], k[b], v[b], cm)
                scale = max(1.0, float(want.abs().max()))
                et = max(et, float((msg[b].double() - want).abs().max()) / scale)
                em = max(em, float((msg[b].double() - t._attention_split_model(q[b] * t.QSCALE, k[b], v[b], cm)).abs().max()) / scale)
            print(f"n {n} bs {bs} qk_scale {qk_scale} nsplit {nsplit}: vs fp64 softmax {et:.2e}   vs fp64 evaluation of the split operands {em:.2e}", flush=True)
PY
  cat "$OUT/att_err.txt" ;;
census_trained)
  timeout 1200 python tools/parity_census.py --families trained_n1000_b1,trained_n5000_b32,trained_kitti_n5000_b16,trained_lomatch_n10000_b8,trained_kitti_n12000_b4,trained_multiway_n20000_b1 --batches 0,1,2,4,8,16,32 > "$OUT/parity_census_trained.txt" 2>&1
  timeout 1200 python tools/parity_census.py --families trained_n1000_b1,trained_n5000_b32,trained_kitti_n5000_b16,trained_lomatch_n10000_b8,trained_kitti_n12000_b4,trained_multiway_n20000_b1 --batches 0,1 --attention-precision fp32 --compat-format f32 --layer-gemm f32 > "$OUT/parity_census_trained_exact_fp32.txt" 2>&1
  grep -E "^trained|strict pass|registration" "$OUT/parity_census_trained.txt" | cut -c1-400 | head -80 ;;
census)
  timeout 1500 python tools/parity_census.py --batches 0,1,2,4,8,16,32 > "$OUT/parity_census.txt" 2>&1
  grep -E "outside the fp32" "$OUT/parity_census.txt" | cut -c1-300 | head -80 ;;
bundle)
  bash tools/gpu_profile_run.sh "$TAG" ${BUNDLE_QUICK:-} ;;
kitti_stage)
  for K in "" "--compat-format f32" "--layer-gemm f32" "--attention-precision fp32 --compat-format f32" "--attention-precision fp32 --compat-format f32 --layer-gemm f32"; do
    echo "== overrides: [$K]" >> "$OUT/kitti_stage.txt"
    timeout 600 python tools/parity_census.py --families kitti_n5000_b16,kitti_n12000_b4 --batches 1,2,8 $K 2>&1 | grep -E "^kitti|outside the fp32|\"pair\"" | cut -c1-420 >> "$OUT/kitti_stage.txt"
  done
  grep -E "==|outside" "$OUT/kitti_stage.txt" | cut -c1-200 ;;
knn_bench)
  python - > "$OUT/knn_bench.txt" 2>&1 <<'PY'
import numpy as np, torch
from pointdsc_amd import ops
for bs, n in ((32, 5000), (16, 5000), (8, 10000), (4, 5000)):
    s = n // 10
    rs = np.random.RandomState(1)
    x = rs.standard_normal((bs, n, 128)).astype(np.float32); x /= np.linalg.norm(x, axis=-1, keepdims=True)
    normed = torch.from_numpy(x).cuda()
    seeds = torch.from_numpy(np.stack([rs.permutation(n)[:s] for _ in range(bs)]).astype(np.int32)).cuda()
    out = {}
    h2 = torch.zeros(bs, n, 32, device="cuda")
    rows, rows_pf, _ = ops.normalize_confidence_pf(normed, h2, torch.zeros(32, device="cuda"), torch.zeros(1, device="cuda"))
    for form, kw in (("matrix", dict(form="matrix")), ("fused_gather", dict(form="fused")), ("fused_pf", dict(form="fused", normed_pf=rows_pf))):
        for _ in range(3): r = ops.knn_seeds(rows, seeds, 40, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): r = ops.knn_seeds(rows, seeds, 40, **kw)
        e1.record(); torch.cuda.synchronize()
        out[form] = (e0.elapsed_time(e1) / 20 * 1e3, r)
    flops = 256.0 * bs * s * n
    print(f"kNN of the seeds, {bs} pairs of N={n}, S={s}, k=40 (incl. the S x N scratch allocation of the wrapper): matrix form {out['matrix'][0]:.1f} us, "
          f"fused (gathered columns) {out['fused_gather'][0]:.1f} us, fused (point-fragment columns) {out['fused_pf'][0]:.1f} us "
          f"({flops / out['fused_pf'][0] / 1e6:.1f} TFLOP/s = {flops / out['fused_pf'][0] / 1e6 / 157.3:.3f} of the fp32-MFMA peak); indices equal: "
          f"{bool(torch.equal(out['matrix'][1], out['fused_gather'][1]) and torch.equal(out['matrix'][1], out['fused_pf'][1]))}")
PY
  cat "$OUT/knn_bench.txt" ;;
match_bench)
  timeout 300 python tools/match_bench.py > "$OUT/match_bench.txt" 2>&1; cat "$OUT/match_bench.txt" ;;
kitti_stage_trained)
  for K in "" "--compat-format f32" "--layer-gemm f32" "--compat-format f32 --layer-gemm f32" "--attention-precision fp32 --compat-format f32"; do
    echo "== overrides: [$K]" >> "$OUT/kitti_stage_trained.txt"
    timeout 600 python tools/parity_census.py --families trained_kitti_n5000_b16 --batches 16 $K 2>&1 | grep -E "^trained|outside the fp32|\"pair\"" | cut -c1-520 >> "$OUT/kitti_stage_trained.txt"
  done
  python - >> "$OUT/kitti_stage_trained.txt" 2>&1 <<'PY'
# logits of pair 50 under each arithmetic against the reference's recorded fp32 logits (census_internals conf32)
import numpy as np, torch
from pointdsc_amd import PointDSC, workloads
name = "trained_kitti_n5000_b16"; w = workloads.WORKLOADS[name]
ix = np.load(f"tests/golden/census_internals_{name}.npz")
model = PointDSC(**w["model"]); model.load_state_dict(workloads.state_dict(name, model.state_dict())); model = model.eval().cuda()
for i in (50, 0, 7):
    one = workloads.batch(name, i, 1)
    data = {k: one[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}; data["testing"] = True
    ref = ix["conf32"][i]
    for att, cf, lg in (("fp16x3", "u16", "h3"), ("fp16x3", "f32", "h3"), ("fp16x3", "u16", "f32"), ("fp16x3", "f32", "f32"), ("fp32", "f32", "f32")):
        model.attention_precision, model.compat_format, model.layer_gemm = att, cf, lg
        with torch.no_grad(): model(data)
        conf = model.workspace_view("conf", 1, w["num_corr"]).cpu().numpy()[: w["num_corr"]]
        top = np.argsort(-ref)[:8]
        print(f"pair {i} attention {att} compat {cf} layer {lg}: max |logit - reference| {np.abs(conf - ref).max():.3e} (relative to max |logit| {np.abs(ref).max():.1f}: {np.abs(conf - ref).max() / np.abs(ref).max():.1e}); "
              f"top-8 by the reference {top.tolist()} here {np.argsort(-conf)[:8].tolist()}")
PY
  cat "$OUT/kitti_stage_trained.txt" | cut -c1-330 ;;
*) echo "unknown step $STEP" ;;
esac
done
ls -la "$OUT" | tail -40
