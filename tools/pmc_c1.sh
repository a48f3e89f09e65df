#!/bin/bash
# tools/pmc_c1.sh [outdir] — SQ counters of k_rxb_tiles in `bench.py --workload c1` (one group of 16 captures: nothing else on the chip)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${1:-c1_pmc}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--workload c1 --steps 2 --warmup 1 --no-cpu --no-verify --no-single --c1-groups 1 --c1-captures 16 ${2:-}"
rm -rf /tmp/pmc1 /tmp/pmc2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc1 -- python "$REPO/bench.py" $ARGS > /tmp/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc2 -- python "$REPO/bench.py" $ARGS > /tmp/pmc2.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_rxb_tiles" in k or "k_tail" in k or "k_rxb" in k:
                res[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_sq.txt", "w") as f:
    for k, cs in res.items():
        f.write(k + "\n")
        for c, v in sorted(cs.items()):
            f.write(f"   {c:28s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}\n")
print(open(out + "/pmc_sq.txt").read())
PY
