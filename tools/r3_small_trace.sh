#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3/small
for nx in 4 102 144; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/small/nx$nx -- python3 tools/size_sweep.py 64 20 $nx > gpurun_out/r3/small/nx$nx.log 2>&1 || exit 1
  cat gpurun_out/r3/small/nx$nx.log | grep "^nx"
  f=$(find gpurun_out/r3/small/nx$nx -name "*kernel_stats.csv" | head -1)
  head -6 "$f" | cut -d, -f1-7 | cut -c1-200
done
