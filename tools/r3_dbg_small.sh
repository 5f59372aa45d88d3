#!/bin/bash
export AMBER_MPN_LIB="$PWD/amber_amd/csrc/variants/libamber_dbg.so"
EXP_GRAPHS=1 EXP_LATENT=64 EXP_STEPS=20 timeout -k 10 200 python tools/exp_dbg.py 0 256 4096 4352 512 2048 1 8 264
