#!/bin/bash
# parity of the current build (forward tests), then A/B timing against variant libraries: tools/r3_ab.sh <variant.so> ...
set -u
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_precision_modes.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/ab_parity.log 2>&1
rc=$?; tail -4 gpurun_out/r3/ab_parity.log; [ $rc -eq 0 ] || exit $rc
AB_ARGS="--no-secondary ${AB_ARGS:-}" bash tools/ab.sh base "$@"
