#!/bin/bash
# full GPU suite on the current build, then the tanh / outer configurations on the config-5 batch: current build vs the
# library of the round's start (row-per-lane kernels for those configurations)
set -u
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread > gpurun_out/r3/var_gputests.log 2>&1
rc=$?; tail -6 gpurun_out/r3/var_gputests.log; [ $rc -eq 0 ] || exit $rc
run() {
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ms/fwd %.3f' % d['ms_per_step'], 'edge %.3f node %.3f' % (k['edge_ms_per_launch'], k['node_ms_per_launch']))
" || exit 1
}
for cfg in "" "--activation tanh" "--residual outer --layer-norm outer" "--residual outer" "--layer-norm outer" "--residual none --layer-norm none --activation relu"; do
  for lib in base amber_amd/csrc/variants/libamber_head.so; do
    if [ "$lib" = "base" ]; then unset AMBER_MPN_LIB; else export AMBER_MPN_LIB="$PWD/$lib"; fi
    echo -n "[$cfg] $lib: "; run $cfg
  done
done
