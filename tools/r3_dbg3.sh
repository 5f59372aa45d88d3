#!/bin/bash
timeout -k 10 120 python tests/diag_modes3.py 64 1 tanh 2 | tail -2
timeout -k 10 120 python tests/diag_modes3.py 128 1 tanh 2 | tail -1
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "tanh_and_outer" --timeout 180 --timeout-method thread 2>&1 | tail -12
exit 0
