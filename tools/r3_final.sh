#!/bin/bash
# round-3 final records: trace + traffic counters, SQ counter passes, the default bench line, the variant configurations
set -u
mkdir -p gpurun_out/r3final
bash tools/profile_round.sh r3_final || exit 1
bash tools/pmc_passes.sh gpurun_out/prof_r3_final/pmc || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r3final/bench.json 2> gpurun_out/r3final/bench.err || { tail -20 gpurun_out/r3final/bench.err; exit 1; }
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --activation tanh > gpurun_out/r3final/bench_tanh.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --residual outer --layer-norm outer > gpurun_out/r3final/bench_outer.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload airfoil1 --no-cpu-baseline > gpurun_out/r3final/bench_airfoil1.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload cap500k --no-cpu-baseline > gpurun_out/r3final/bench_cap500k.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --train --no-cpu-baseline > gpurun_out/r3final/bench_train.json 2>/dev/null || exit 1
ls -la gpurun_out/r3final
