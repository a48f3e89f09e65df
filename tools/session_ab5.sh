for rep in 1 2; do for w in 48 96 192; do
  echo -n "placement + SWPC=$w rep $rep: "
  LSDR_MFMA_SWPC=$w timeout 200 python bench.py --steps 20 --warmup 5 --no-more --no-cpu 2>/dev/null | python tools/bench_brief.py | head -1
done; done
for w in 4 48; do
  echo "== c2_offset SWPC=$w"
  LSDR_MFMA_SWPC=$w timeout 300 python tools/more_one.py c2_offset 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('c2_offset '):
        j=json.loads(l.split(' ',1)[1]); print('c2_offset', j['value'], j['pass'], j['roofline']['avg_launch_ms'])
"
done
for w in 32 64; do
  echo "== anf1 NF_WPC=$w"
  LSDR_NF_WPC=$w LSDR_MORE_VERIFY=1 timeout 300 python tools/more_one.py anf1 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('anf1 '):
        j=json.loads(l.split(' ',1)[1]); print('anf1', j['value'], j['pass'], j['roofline']['avg_launch_ms'])
"
done
