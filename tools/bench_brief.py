"""print the key numbers of a bench.py JSON line read from stdin: value, ms/step, roofline frac, kernel ms, tile stats"""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"],
      d["config"].get("rx_tile"), d["config"].get("rx_tiles"))
