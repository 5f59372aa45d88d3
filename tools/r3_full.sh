#!/bin/bash
# full GPU suite + headline bench + the non-default configurations' timings
set -u
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread > gpurun_out/r3/full_gputests.log 2>&1
rc=$?; tail -12 gpurun_out/r3/full_gputests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || { tail -20 gpurun_out/r3/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench_default.json"))
k=d["kernels"]; s=d["secondary"]
print("headline ms/fwd %.3f value %.1f M edges/s whole-forward frac %.3f; edge steady %.3f first %.3f node steady %.3f last %.3f; roofline frac %.3f" % (
    d["ms_per_step"], d["value"]/1e6, d["hbm_roofline_frac_whole_forward"], k["edge_ms_per_launch_steady_state"], k["edge_ms_first_launch"],
    k["node_ms_per_launch_steady_state"], k["node_ms_last_launch"], d["roofline"]["frac"]))
print("airfoil1 %.4f ms  cap500k %.4f ms  train %.3f ms/step" % (s["airfoil1"]["ms_per_forward_pipelined"], s["cap500k"]["ms_per_forward_pipelined"], s["train_config3"]["ms_per_step"]))
PY
for cfg in "--activation tanh" "--residual outer --layer-norm outer" "--residual outer" "--layer-norm outer"; do
  echo -n "[$cfg]: "
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ms/fwd %.3f' % d['ms_per_step'], 'edge %.3f node %.3f' % (k['edge_ms_per_launch'], k['node_ms_per_launch']))
" || exit 1
done
