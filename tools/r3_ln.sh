#!/bin/bash
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 180 --timeout-method thread 2>&1 | tail -3 || exit 1
AB_ARGS=--no-secondary bash tools/ab.sh base amber_amd/csrc/variants/libamber_oldln.so || exit 1
AB_ARGS="--workload cap500k" bash tools/ab.sh base amber_amd/csrc/variants/libamber_oldln.so
