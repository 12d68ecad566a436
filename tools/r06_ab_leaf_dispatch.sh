mkdir -p gpurun_out/r06e
export POINTDSC_HIP_LIB=$PWD/pointdsc_amd/libpointdsc_hip_exp.so
run() { # name, env..., args
  local n=$1; shift
  env "$@" > /dev/null 2>&1
}
for rep in 1 2; do
 python bench.py --no-cpu-baseline --extra off --sustain-seconds 1.5 > gpurun_out/r06e/canon_$rep.json 2>/dev/null
 python bench.py --no-cpu-baseline --extra off --sustain-seconds 1.5 --att-leaves per_launch > gpurun_out/r06e/perlaunch2_$rep.json 2>/dev/null
 PDSC_ATT_SPLIT_NS=4 python bench.py --no-cpu-baseline --extra off --sustain-seconds 1.5 --att-leaves per_launch --no-check > gpurun_out/r06e/perlaunch4_$rep.json 2>/dev/null
 PDSC_ATT_SPLIT_NS=1 python bench.py --no-cpu-baseline --extra off --sustain-seconds 1.5 --att-leaves per_launch --no-check > gpurun_out/r06e/perlaunch1_$rep.json 2>/dev/null
 python bench.py --no-cpu-baseline --extra off --sustain-seconds 1.5 --att-leaves 2 > gpurun_out/r06e/leaves2_$rep.json 2>/dev/null
done
unset POINTDSC_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=line -s -k "stage_decisions or leaf_count_classes or layer_fused_frag_h3 or pipelined_kernel_is_bit" 2>&1 | grep -v "^$" | grep "STAGE-CENSUS\|passed\|failed\|Error\|assert" | cut -c1-1200 > gpurun_out/r06e/new_tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06e/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["sustained"]["value"], "att", d["roofline"]["avg_launch_ms"], "lay", d["roofline_layer"]["avg_launch_ms"], d["config"]["attention_plan"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r06e/new_tests.txt
