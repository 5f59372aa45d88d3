#!/bin/bash
mkdir -p gpurun_out/r3final
timeout -k 10 400 python bench.py > gpurun_out/r3final/bench.json 2> gpurun_out/r3final/bench.err || { tail -20 gpurun_out/r3final/bench.err; exit 1; }
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --activation tanh > gpurun_out/r3final/bench_tanh.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --residual outer --layer-norm outer > gpurun_out/r3final/bench_outer.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload airfoil1 --no-cpu-baseline > gpurun_out/r3final/bench_airfoil1.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload cap500k --no-cpu-baseline > gpurun_out/r3final/bench_cap500k.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --train --no-cpu-baseline > gpurun_out/r3final/bench_train.json 2>/dev/null || exit 1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3final/bench.json')); k=d['kernels']; r=d['roofline']
print('value %.1f ms %.3f edge %.3f node %.3f traffic %s'%(d['value']/1e6, d['ms_per_step'], k['edge_ms_per_launch_steady_state'], k['node_ms_per_launch_steady_state'], r['traffic']))
PY
