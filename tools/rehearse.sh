#!/bin/bash
# One-GPU rehearsals of the multi-rank bench paths (gloo, both ranks on cuda:0): the JSON lines go to gpurun_out/rehearse_*.json.
# usage: tools/rehearse.sh [ranks]      (not a scaling measurement: see bench.py --rehearse-on-one-gpu)
set -u
N=${1:-2}
run() {  # name, extra args...
  local name=$1; shift
  AMBER_BENCH_WATCHDOG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
    --master-port $((29500 + RANDOM % 500)) bench.py --gpus "$N" --steps 5 --warmup 2 --rehearse-on-one-gpu "$@" \
    > "gpurun_out/rehearse_$name.json" 2> "gpurun_out/rehearse_$name.err"
  echo "$name exit $?"
}
run forward --graphs 20 || exit 1
run train --train || exit 1
