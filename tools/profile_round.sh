#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats over bench.py, then FETCH_SIZE and WRITE_SIZE counter
# passes (each in a run of its own) over a single forward.  usage: tools/profile_round.sh <tag> [extra bench.py arguments]
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace exit $?"
if [ $# -eq 0 ]; then
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/prof_forward.py 100 1 > "$OUT/fetch.log" 2>&1
  echo "fetch exit $?"
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/prof_forward.py 100 1 > "$OUT/write.log" 2>&1
  echo "write exit $?"
fi
