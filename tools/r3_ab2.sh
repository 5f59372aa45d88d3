#!/bin/bash
# parity of the current build, then A/B on the one-mesh workload, the size-capped batch and the headline
set -u
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_precision_modes.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/ab_parity.log 2>&1
rc=$?; tail -3 gpurun_out/r3/ab_parity.log; [ $rc -eq 0 ] || exit $rc
for wl in airfoil1 cap500k; do
  echo "== $wl"
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = "base" ]; then unset AMBER_MPN_LIB; else export AMBER_MPN_LIB="$PWD/$lib"; fi
      timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$lib', 'ms/fwd %.4f' % d['ms_per_step'], 'edge %.4f node %.4f embed %.4f' % (k['edge_ms_per_launch'], k['node_ms_per_launch'], k['embed_ms_per_forward']))
" || exit 1
    done
  done
done
echo "== config5"
AB_ARGS="--no-secondary" bash tools/ab.sh "$@"
