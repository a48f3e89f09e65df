"""debug: info text of one leandvb_bench run (GPU apps or --ref) → gpurun_out/info_{gpu,ref}_{case}.txt
usage: bench_one.py [--ref] case   (case: sps12 | sps4_viterbi_rrc | sps12_hs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import leandvb_bench as lb
CASES = {"sps12": ("6/5", 18, "", 700), "sps4_viterbi_rrc": ("4", 5.5, "--viterbi --sampler rrc", 500), "sps12_hs": ("6/5", 15, "--u8 --hs", 700)}
ref = "--ref" in sys.argv
names = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
if not ref:
    lb.RX_EXTRA = "--buf-factor 4"
for n in names:
    text, ts = lb.run_pipeline(*CASES[n], ref)
    open(os.path.join(lb.ROOT, "gpurun_out", f"info_{'ref' if ref else 'gpu'}_{n}.txt"), "w").write(text)
    print(n, len(text.splitlines()), len(ts) // 188)
