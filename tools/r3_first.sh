#!/bin/bash
# round 3, first GPU call: full GPU suite, headline bench with the secondary figures, self-launched 2-rank rehearsal
set -u
mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/gputests.log 2>&1
rc=$?; tail -5 gpurun_out/r3/gputests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err || { tail -20 gpurun_out/r3/bench_default.err; exit 1; }
cat gpurun_out/r3/bench_default.json
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --graphs 20 --steps 5 --warmup 2 > gpurun_out/r3/rehearse_selflaunch_forward.json 2> gpurun_out/r3/rehearse_selflaunch_forward.err || { tail -20 gpurun_out/r3/rehearse_selflaunch_forward.err; exit 1; }
cat gpurun_out/r3/rehearse_selflaunch_forward.json
timeout -k 10 300 python bench.py --gpus 2 --rehearse-on-one-gpu --train --steps 5 --warmup 2 > gpurun_out/r3/rehearse_selflaunch_train.json 2> gpurun_out/r3/rehearse_selflaunch_train.err || { tail -20 gpurun_out/r3/rehearse_selflaunch_train.err; exit 1; }
cat gpurun_out/r3/rehearse_selflaunch_train.json
