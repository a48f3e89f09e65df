for rep in 1 2 3; do for w in 3 48; do
  echo -n "placement + SWPC=$w rep $rep: "
  LSDR_MFMA_SWPC=$w timeout 200 python bench.py --steps 20 --warmup 5 --no-more --no-cpu 2>/dev/null | python tools/bench_brief.py | head -1
  python -c "
import json; j=json.load(open('bench_full.json')); p=j['config']['buffer_placement']; print('    in', p['filter_launch_ms_by_input_buffer'], 'dec', p['filter_launch_ms_by_decimated_buffer'])"
done; done
