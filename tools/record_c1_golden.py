#!/usr/bin/env python3
"""tools/record_c1_golden.py BENCH_C1.json — turn the `verified` object of a `bench.py --workload c1` line that was produced NEXT TO the
reference binary (oracle/_ref/leandvb decoded every capture's IQ) into tests/golden/c1_ts.json: per capture the SHA-256 of its IQ, of the
reference's TS and of this path's TS, so that a machine without the reference binary can still pin its TS to a reference-checked one."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v = j["verified"]
assert v["pass"] and "oracle/_ref/leandvb" in v["checker"], "record only from a run the reference binary checked"
caps = []
for c in v["per_capture"]:
    assert c["equal_to_reference_after_acquisition"]
    caps.append({k: c[k] for k in ("samples", "seed", "first_packet", "ts_packets", "ref_packets", "iq_sha256", "ref_ts_sha256", "ts_sha256",
                                   "equal_to_reference_after_acquisition", "whole_ts_identical", "compared")})
out = dict(_comment="bench_c1.py captures (device-generated: tx.hip / chan.hip, srand48(seed) noise) decoded by oracle/_ref/leandvb "
                    "--u8 -f 2400e3 --sr 2000e3 --cr 1/2 --anf 0 on the GPU box's host; recorded by tools/record_c1_golden.py",
           reference_flags=v["checker"], rx_tile=j["config"]["rx_tile"], captures=caps)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c1_ts.json"), "w"), indent=1)
print(f"recorded {len(caps)} captures")
