#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/r3/train_trace; mkdir -p gpurun_out/r3/train_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/train_trace -- python3 bench.py --train --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3/train_trace/bench.json 2> gpurun_out/r3/train_trace/err.log || { tail -5 gpurun_out/r3/train_trace/err.log; exit 1; }
cat gpurun_out/r3/train_trace/bench.json | cut -c1-300
f=$(find gpurun_out/r3/train_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'per step (25 steps)', tot/1e6/25)
for r in rows[:28]:
    print('%-70s calls %6s  total %.2f ms  avg %.1f us  %.1f%%'%(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
