#!/bin/bash
mkdir -p gpurun_out/c19
export TMPDIR=/tmp
# the reference's sensitivity benchmark: (1) reference app SOURCES on the GPU headers, reference pipe sizes (exact modes);
# (2) this repo's apps in the throughput modes (--tiled: receiver tiles + scan notch), all four series incl. the rrc sampler
python - <<'PY' > gpurun_out/c19/refgraph_exact_mi355x.txt 2>gpurun_out/c19/refgraph.err
import sys, os
sys.path.insert(0, "tools")
import leandvb_bench as lb
for name in ("1.2sps", "4sps-viterbi-rrc", "1.2sps-hs"):
    ratio, snrs, flags = lb.SERIES[name]
    print(f"# {name}.")
    for snr in snrs:
        text, _ = lb.run_pipeline(ratio, snr, flags, 1500, ref="graph")
        r = lb.parse_info(text, 500)
        rr = eval(ratio) if "/" in ratio else float(ratio)
        rxsnr = lb.commands(ratio, snr, flags)[3]
        print(f"refgraph {rr:.2f} {rxsnr:.2f} " + ("no-lock" if r is None else f"{r['cnr']:g} {r['ss']:g} {r['mer']:g} {r['vbermin']:.6f} {r['vbermax']:.6f}"), flush=True)
PY
python tools/leandvb_bench.py --packets 1500 --min-packets 500 --rx-extra "--tiled --buf-factor 64" 1.2sps 4sps-viterbi-rrc 1.2sps-hs 1.2sps-viterbi > gpurun_out/c19/leansdr_amd_tiled_mi355x.txt 2> gpurun_out/c19/tiled.err
cat gpurun_out/c19/refgraph_exact_mi355x.txt gpurun_out/c19/leansdr_amd_tiled_mi355x.txt
