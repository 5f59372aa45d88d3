#!/bin/bash
set -u
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_precision_modes.py tests/test_gpu_training.py -m gpu -q --timeout 180 --timeout-method thread > gpurun_out/r3/tanh_tests.log 2>&1
rc=$?; tail -4 gpurun_out/r3/tanh_tests.log; [ $rc -eq 0 ] || exit $rc
for cfg in "--activation tanh" ""; do
  echo -n "[$cfg]: "
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ms/fwd %.3f' % d['ms_per_step'], 'edge %.3f node %.3f' % (k['edge_ms_per_launch'], k['node_ms_per_launch']))
" || exit 1
done
