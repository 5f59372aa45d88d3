#!/bin/bash
for lib in base amber_amd/csrc/variants/libamber_head.so amber_amd/csrc/variants/libamber_noearly.so; do
  if [ "$lib" = "base" ]; then unset AMBER_MPN_LIB; else export AMBER_MPN_LIB="$PWD/$lib"; fi
  echo "== $lib"
  timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "tanh_and_outer and f16x2" --timeout 180 --timeout-method thread 2>&1 | grep "passed\|failed\|FAILED"
done
exit 0
