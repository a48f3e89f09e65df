#!/usr/bin/env python3
"""tools/clk_trace.py OUT.csv -- CMD...   run CMD while sampling the GPU's shader clock and power every 20 ms.

Sources, in order of preference: the amdgpu hwmon files of card 0 (freq1_input = sclk in Hz, power1_average / power1_input in
microwatts; a file read costs microseconds), else `amd-smi metric --clock --power --json` (slow: one sample per call).
Writes `t_seconds,sclk_mhz,power_w` rows and prints a summary (idle / busy medians, min and max while busy)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def hwmon_paths():
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if not hw:
            continue
        f = os.path.join(hw[0], "freq1_input")
        p = [q for q in (os.path.join(hw[0], "power1_average"), os.path.join(hw[0], "power1_input")) if os.path.exists(q)]
        if os.path.exists(f):
            return f, (p[0] if p else None)
    return None, None


def read_num(path):
    try:
        with open(path) as fh:
            return float(fh.read().strip())
    except (OSError, ValueError):
        return float("nan")


def smi_sample():
    try:
        r = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=10)
        j = json.loads(r.stdout.decode())
        g = j[0] if isinstance(j, list) else j.get("gpu_data", [j])[0]
        clk = g.get("clock", {})
        gfx = [v for k, v in clk.items() if k.startswith("gfx")]
        mhz = max(float(v["clk"]["value"] if isinstance(v.get("clk"), dict) else v.get("clk", 0)) for v in gfx) if gfx else float("nan")
        pw = g.get("power", {}).get("socket_power", {})
        watts = float(pw["value"] if isinstance(pw, dict) else pw)
        return mhz, watts
    except Exception:
        return float("nan"), float("nan")


def main():
    out, sep = sys.argv[1], sys.argv.index("--")
    cmd = sys.argv[sep + 1:]
    fclk, fpow = hwmon_paths()
    rows, stop = [], threading.Event()

    def sampler():
        t0 = time.perf_counter()
        while not stop.is_set():
            if fclk:
                mhz, w = read_num(fclk) / 1e6, (read_num(fpow) / 1e6 if fpow else float("nan"))
                rows.append((time.perf_counter() - t0, mhz, w))
                time.sleep(0.02)
            else:
                mhz, w = smi_sample()
                rows.append((time.perf_counter() - t0, mhz, w))

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=15)
    with open(out, "w") as fh:
        fh.write("t_seconds,sclk_mhz,power_w\n")
        for r in rows:
            fh.write("%.3f,%.0f,%.1f\n" % r)
    ws = sorted(r[2] for r in rows if r[2] == r[2])
    if ws:
        thr = ws[0] + 0.5 * (ws[-1] - ws[0])
        busy = [r for r in rows if r[2] >= thr]
        idle = [r for r in rows if r[2] < thr]
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
        print("clk_trace: source=%s samples=%d busy=%d | sclk busy median %.0f MHz (min %.0f, max %.0f), idle median %.0f MHz | power busy median %.0f W (max %.0f), idle %.0f W"
              % (fclk or "amd-smi", len(rows), len(busy), med([r[1] for r in busy]), min([r[1] for r in busy] or [float("nan")]),
                 max([r[1] for r in busy] or [float("nan")]), med([r[1] for r in idle]), med([r[2] for r in busy]), ws[-1], med([r[2] for r in idle])))
    else:
        print("clk_trace: no power readings (source=%s, %d samples)" % (fclk or "amd-smi", len(rows)))
    sys.exit(rc)


if __name__ == "__main__":
    main()
