#!/bin/bash
# The fault of DESIGN.md section 5.3 on the current sources: the general translation unit built WITHOUT the pkfix rewrite
# against the product build, same forwards.  First, on the CPU:
#   AMBER_NO_PKFIX=1 tools/build_variant.sh nofix mpn_f2_general.hip     (tools/check_isa.py on it: 8 kernels hold the form)
# then on the GPU box: tools/repro/without_pkfix.sh
# Round 3 on an MI355X: 14 and 68 distinct results in 150 forwards (1 and 2 steps) without the rewrite, 1 and 1 with it;
# with EVERY translation unit built without it: 150 of 150 distinct.
for lib in amber_amd/csrc/variants/libamber_nofix.so base; do
  if [ "$lib" = base ]; then unset AMBER_MPN_LIB; else export AMBER_MPN_LIB="$PWD/$lib"; fi
  echo "== $lib"
  timeout -k 10 200 python tools/run_to_run.py 64 relu outer inner 2 150 2>&1 | grep -v "^   run\|amdgpu.ids" || exit 1
done
