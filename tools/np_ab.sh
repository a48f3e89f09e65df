# tools/np_ab.sh — A/B of the stream kernel's rows per wave tile (LSDR_MFMA_NP / _NP_CP / LSDR_NF_NP) on one box
mkdir -p gpurun_out/np
for i in 1 2; do
  for cfg in "8 96" "4 96" "4 192" "4 48"; do
    set -- $cfg
    echo "== NP $1 SWPC $2 round $i"
    LSDR_MFMA_NP=$1 LSDR_MFMA_SWPC=$2 timeout 200 python bench.py --no-more --no-cpu --no-verify --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config'].get('buffer_placement'))"
  done
done > gpurun_out/np/headline.txt 2>&1
for cfg in "4 64" "4 128" "4 32"; do
  set -- $cfg
  echo "== NF_NP $1 NF_WPC $2"
  LSDR_NF_NP=$1 LSDR_NF_WPC=$2 timeout 200 python tools/more_one.py anf1 2>&1 | cut -c1-120
done > gpurun_out/np/anf1.txt 2>&1
cat gpurun_out/np/headline.txt gpurun_out/np/anf1.txt
