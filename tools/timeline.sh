#!/bin/bash
# kernel timeline of a few bench steps (rocprofv3 --kernel-trace): gpurun_out/timeline.csv (name, start, end in ns)
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python "$REPO/bench.py" --steps 8 --warmup 2 --no-cpu "$@" > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" "$REPO/gpurun_out/timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    for r in rows:
        o.write(f"{r['Kernel_Name'][:60].replace(',', ';')},{r['Start_Timestamp']},{r['End_Timestamp']},{r.get('Queue_Id','')}\n")
print(len(rows), "kernels")
PY
