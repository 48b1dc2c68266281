#!/bin/bash
# rocprofv3 kernel trace of the training bench -> gpurun_out/$1/train_kernel_stats.csv (+ the bench line);  gpurun -- bash profiles/scripts/prof_train_trace.sh r6x
export TMPDIR=/tmp
O=gpurun_out/${1:-r6t}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline > $O/train_prof.json 2> $O/train_prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; rm -rf $O/prof
python bench.py --workload train --steps 30 --warmup 10 --no-cpu-baseline > $O/train.json 2> $O/train.err
python - <<P
import csv, json
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
n=[int(r['Calls']) for r in rows if 'k_adam' in r['Name']][0]+2
print('kernel us/step', round(sum(int(r['TotalDurationNs']) for r in rows)/n/1e3), 'launches/step', round(sum(int(r['Calls']) for r in rows)/n,1))
for r in rows[:28]: print(' ', r['Name'][:70].ljust(70), round(int(r['Calls'])/n,1), round(float(r['AverageNs'])/1e3,2))
d=json.load(open("$O/train.json")); print('graphed ms', d['graphed']['ms_per_step'], 'launches', d['launches_per_step'])
P
