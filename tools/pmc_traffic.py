#!/usr/bin/env python3
"""tools/pmc_traffic.py OUT.json -- CMD...   HBM-side traffic per kernel of CMD from two rocprofv3 PMC passes.

One counter per pass (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md §rocprofv3 PMC slots), no
tracing options next to --pmc.  Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of a wide
coalesced stream at 64 bytes (§HBM of the same guide): `fetch_x2` doubles it — the calibrated case is 16 B per lane streaming
(global_load_dwordx4, buffer_load … lds); for other access shapes the raw figure is given next to it.  Kernels are serialised
while counters are collected, so figures are per kernel, not per overlapped pipeline.  GPU box only."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def one_pass(counter, cmd):
    d = f"/tmp/pmc_{counter}_{os.getpid()}"
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            agg[k][0] += 1
            agg[k][1] += float(row["Counter_Value"])
    shutil.rmtree(d, ignore_errors=True)
    return agg, r.returncode, r.stdout.decode()[-2000:]


def main():
    out, sep = sys.argv[1], sys.argv.index("--")
    cmd = sys.argv[sep + 1:]
    fetch, rc1, log1 = one_pass("FETCH_SIZE", cmd)
    write, rc2, log2 = one_pass("WRITE_SIZE", cmd)
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1])):
        f, w = fetch.get(k, [0, 0.0]), write.get(k, [0, 0.0])
        rows.append(dict(kernel=k[:160], launches=max(f[0], w[0]), fetch_kib=f[1], write_kib=w[1],
                         bytes_fetch_x2_plus_write=int((2 * f[1] + w[1]) * 1024), bytes_fetch_raw_plus_write=int((f[1] + w[1]) * 1024)))
    res = dict(command=" ".join(cmd), returncodes=[rc1, rc2], kernels=rows,
               total_bytes_fetch_x2_plus_write=sum(r["bytes_fetch_x2_plus_write"] for r in rows),
               total_bytes_fetch_raw_plus_write=sum(r["bytes_fetch_raw_plus_write"] for r in rows))
    json.dump(res, open(out, "w"), indent=1)
    for r in rows[:16]:
        print(f"{r['kernel'][:70]:72s} x{r['launches']:5d}  fetch {r['fetch_kib']/1024:10.1f} MiB  write {r['write_kib']/1024:10.1f} MiB")
    print("total (fetch x2 + write): %.1f MiB; (fetch raw + write): %.1f MiB" % (res["total_bytes_fetch_x2_plus_write"] / 2**20, res["total_bytes_fetch_raw_plus_write"] / 2**20))
    if rc1 or rc2:
        print(log1[-600:], log2[-600:])


if __name__ == "__main__":
    main()
