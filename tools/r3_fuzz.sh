#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 500 python -u tests/fuzz_parity.py --cases 200 --seed 51 --keep-going > gpurun_out/r3/fuzz_fwd.log 2>&1; rc1=$?
tail -3 gpurun_out/r3/fuzz_fwd.log
[ $rc1 -eq 124 -o $rc1 -eq 137 ] && exit $rc1
timeout -k 10 500 python -u tests/fuzz_parity.py --cases 100 --seed 52 --grads --keep-going > gpurun_out/r3/fuzz_grads.log 2>&1; rc2=$?
tail -3 gpurun_out/r3/fuzz_grads.log
exit $((rc1 + rc2))
