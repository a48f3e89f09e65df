#!/bin/bash
# tools/headline_repeat.sh [n] ["extra bench.py arguments"] — the C2 headline n times back to back (processes of one box): value, filter launch, the unplaced figure, what lsdr_arena_place saw
cd "$(dirname "$0")/.."
for r in $(seq 1 ${1:-4}); do
  timeout 300 python bench.py --no-more --no-cpu $2 2>/dev/null > /dev/null
  python - <<PY
import json
d=json.load(open("bench_full.json"))
bp=d["config"]["buffer_placement"] or {}
ti=bp.get("filter_launch_ms_by_input_window",[]); td=[v for l in bp.get("filter_launch_ms_by_decimated_window",[]) for v in l] or [0]
print("run $r:", d["value"], "frac", d["roofline"]["frac"], "launch ms", d["roofline"]["avg_launch_ms"], "unplaced", (d.get("unplaced") or {}).get("value"), (d.get("unplaced") or {}).get("frac"),
      "| input windows: best %.4f median %.4f worst %.4f of %d | decimated windows: best %.4f median %.4f worst %.4f of %d" % (min(ti), sorted(ti)[len(ti)//2], max(ti), len(ti), min(td), sorted(td)[len(td)//2], max(td), len(td)), bp.get("input_windows_paired"), bp.get("chosen_pair"), [(e["set"][:14], e["ms"]) for e in bp.get("pipeline_ms_per_batch_by_output_set", [])], bp.get("output_set_in_use"), "verified", d["verified"]["pass"])
PY
done
