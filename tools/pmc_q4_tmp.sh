#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_IFETCH_LEVEL"; do
  rm -rf /tmp/pq
  rocprofv3 --pmc $set --output-format csv -d /tmp/pq -- python /root/repo/tools/vit_alone.py ${N:-16} > /dev/null 2>&1
  python3 - <<'E'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'viterbi' not in row['Kernel_Name'] or 'verify' in row['Kernel_Name']: continue
        k = row['Counter_Name']; agg[k][0] += 1; agg[k][1] += float(row['Counter_Value'])
for k, (n, v) in agg.items(): print(f"{k:24s} launches {n:4d}  per launch {v/n:16.1f}")
E
done
