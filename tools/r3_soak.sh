#!/bin/bash
# soak: the GPU suite three times, then the bit-equality detectors twenty times
mkdir -p gpurun_out/r3
for k in 1 2 3; do
  timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 180 --timeout-method thread > gpurun_out/r3/soak_$k.log 2>&1 || { tail -15 gpurun_out/r3/soak_$k.log; exit 1; }
  tail -1 gpurun_out/r3/soak_$k.log
done
for k in $(seq 1 20); do
  timeout -k 10 120 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forty or bit" --timeout 100 --timeout-method thread > gpurun_out/r3/soak_bits.log 2>&1 || { tail -15 gpurun_out/r3/soak_bits.log; exit 1; }
done
tail -1 gpurun_out/r3/soak_bits.log
echo "soak ok"
