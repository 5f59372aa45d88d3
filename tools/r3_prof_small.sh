#!/bin/bash
# kernel traces of the training step (config-3 shape) and the one-mesh forward
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof_train -- python3 bench.py --train --steps 20 --warmup 5 > gpurun_out/r3/prof_train.json 2> gpurun_out/r3/prof_train.err
echo "train exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/prof_airfoil -- python3 bench.py --workload airfoil1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3/prof_airfoil.json 2> gpurun_out/r3/prof_airfoil.err
echo "airfoil exit $?"
for d in prof_train prof_airfoil; do
  f=$(find gpurun_out/r3/$d -name "*_kernel_stats.csv" | head -1)
  echo "== $d"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:95]:<95} {int(r['Calls']):>6} {float(r['TotalDurationNs'])/1e6:>9.3f} {float(r['AverageNs'])/1e3:>9.2f}us {float(r['Percentage']):>6.2f}")
PY
done
