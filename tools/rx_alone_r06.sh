cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
MEASURE_LIB=$PWD/tools/variants/liblsdr_hip_measure.so
LSDR_HIP_LIB=$MEASURE_LIB LSDR_FIR_SKIP=1 bash tools/timeline.sh --no-more --no-verify --batches-per-step 8 > /dev/null 2>&1
python - > gpurun_out/r06_prof_rx_alone.txt <<'PY'
import re, collections
d = collections.defaultdict(list)
for ln in open("gpurun_out/timeline.csv"):
    p = ln.split(",")
    m = re.search(r"(k_rx\w+|k_fir\w+)", p[0])
    if m: d[m.group(1)].append(int(p[2]) - int(p[1]))
print("receiver kernels of the C2 pipeline with the filter launches skipped (measure build, LSDR_FIR_SKIP=1): per batch")
for k, v in d.items():
    v = v[len(v) // 2:]
    print(f"  {k:28s} n={len(v):4d} mean {sum(v) / len(v) / 1e3:8.1f} us")
PY
cat gpurun_out/r06_prof_rx_alone.txt
for w in 256 128; do
  timeout 300 python bench.py --no-more --no-cpu --tile-warmup $w 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warmup $w', j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['verified'])"
done
