#!/bin/bash
set -u
mkdir -p gpurun_out/r3
AMBER_MPN_LIB=$PWD/amber_amd/csrc/variants/libamber_stamps.so timeout -k 10 200 python bench.py --workload airfoil1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3/stamps_airfoil.json 2> gpurun_out/r3/stamps_airfoil.err
echo "exit $?"
grep "stamps" gpurun_out/r3/stamps_airfoil.err | sort | uniq -c | sort -rn | head -5
grep "node stamps L=64 HEAD=0" gpurun_out/r3/stamps_airfoil.err | tail -3
grep "\[stamps L=64 EMB=0" gpurun_out/r3/stamps_airfoil.err | tail -3
