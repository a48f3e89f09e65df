for rep in 1 2 3 4 5; do
  echo -n "placement search rep $rep: "
  timeout 200 python bench.py --steps 20 --warmup 5 --no-more --no-cpu 2>/dev/null | python tools/bench_brief.py | head -1
  python -c "
import json; j=json.load(open('bench_full.json')); print('   ', j['config']['buffer_placement'])"
done
for rep in 1 2 3; do
  echo -n "no search rep $rep: "
  LSDR_BENCH_PLACEMENT=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-more --no-cpu 2>/dev/null | python tools/bench_brief.py | head -1
done
