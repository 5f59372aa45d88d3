#!/bin/bash
# A/B timing of library builds in ONE GPU call: tools/ab.sh <lib1.so> <lib2.so> ...  (paths relative to the repo root; AB_ARGS = extra bench.py arguments;
# "base" = the default library).  Prints ms/forward and per-kernel-class ms for each, twice, interleaved.
set -u
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then unset AMBER_MPN_LIB; else export AMBER_MPN_LIB="$PWD/$lib"; fi
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$lib', 'ms/fwd %.3f' % d['ms_per_step'], 'edge %.3f node %.3f embed %.3f' % (k['edge_ms_per_launch'], k['node_ms_per_launch'], k['embed_ms_per_forward']))
" || exit 1
  done
done
