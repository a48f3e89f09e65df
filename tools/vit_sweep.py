"""c3 chain with viterbi tile length / warm-up variants (LSDR_VIT_TL, LSDR_VIT_WARM are read once per process)."""
import os, subprocess, sys
for tl in ("0", "8", "4", "2", "1"):
    for warm in ("4", "2"):
        env = dict(os.environ, LSDR_VIT_TL=tl, LSDR_VIT_WARM=warm)
        r = subprocess.run([sys.executable, "tools/more_one.py", "c3"], env=env, capture_output=True, text=True)
        import json
        line = [l for l in r.stdout.splitlines() if l.startswith("c3 ")]
        if not line:
            print(tl, warm, "FAILED", r.stderr[-300:]); continue
        d = json.loads(line[0][3:])
        print("TL", tl, "warm", warm, d["value"], d["host_seconds_per_stage"], d["viterbi"], d["ts_check"]["pass"], flush=True)
