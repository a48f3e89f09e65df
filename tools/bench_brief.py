"""print the key numbers of a bench.py JSON line read from stdin: the headline and the trailing `summary` table"""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "MS/s", d["ms_per_step"], "ms/step; roofline frac", r.get("frac"), "kernel ms", r.get("avg_launch_ms"),
      "verified", (d.get("verified") or {}).get("pass"))
for k, v in (d.get("summary") or {}).items():
    print(f"  {k:20s} {v}")
