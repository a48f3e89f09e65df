#!/bin/bash
# tools/vit_q4_report.sh OUT.txt — viterbi_sync alone (GPU box): lane = state kernel against k_viterbi_q4 at several input lengths,
# and the SQ counters of k_viterbi_q4 that show what bounds it (separate rocprofv3 --pmc passes, no tracing next to them).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "${1:-$ROOT/gpurun_out/vit_q4.txt}")
mkdir -p "$(dirname "$OUT")"
{
  echo "== viterbi_sync alone (tools/vit_alone.py N: N Mi symbols resident, ms per lsdr_viterbi_run call)"
  for n in 4 16 32 64; do
    for k in "auto" "LSDR_VIT_LANE=1" "LSDR_VIT_Q4=1"; do
      echo "-- QPSK 1/2, $n Mi symbols, kernel: $k"; if [ "$k" = auto ]; then python $ROOT/tools/vit_alone.py $n | tail -1; else env $k python $ROOT/tools/vit_alone.py $n | tail -1; fi
      echo "-- 8PSK 2/3, $n Mi symbols, kernel: $k"; if [ "$k" = auto ]; then LSDR_VA_8PSK=1 python $ROOT/tools/vit_alone.py $n | tail -1; else env $k LSDR_VA_8PSK=1 python $ROOT/tools/vit_alone.py $n | tail -1; fi
    done
  done
  cd /tmp && export TMPDIR=/tmp
  for mode in "" "LSDR_VA_8PSK=1"; do
    for tl in 32 8; do
      echo "== SQ counters per k_viterbi_q4 launch, 32 Mi symbols, ${mode:-QPSK 1/2}, tile length $tl chunks (LSDR_VIT_TL=$tl)"
      for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY"; do
        rm -rf /tmp/pq
        env $mode LSDR_VIT_Q4=1 LSDR_VIT_TL=$tl rocprofv3 --pmc $set --output-format csv -d /tmp/pq -- python $ROOT/tools/vit_alone.py 32 > /dev/null 2>&1
        python3 - <<'E'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'viterbi_q4' not in row['Kernel_Name']: continue
        agg[row['Counter_Name']][0] += 1; agg[row['Counter_Name']][1] += float(row['Counter_Value'])
for k, (n, v) in sorted(agg.items()): print(f"  {k:22s} {v/n:16.0f}   (mean of {n} launches)")
E
      done
    done
  done
  echo "(SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1 means 4 cycles per wave64 VALU instruction;"
  echo " SQ_WAVE_CYCLES * 4 / SQ_WAVES = cycles a wavefront lives; GRBM_GUI_ACTIVE / 8 = the launch in shader-clock cycles.)"
} > "$OUT" 2>&1
tail -5 "$OUT"
