for b in 64 32 16 8; do LSDR_MORE_BATCH=$b timeout 100 python tools/more_one.py anf1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('anf1 '):
        d = json.loads(l[5:]); print($b, d['value'], d['seconds'], d['fir_filter_avg_launch_ms'], d['roofline']['avg_launch_ms'])
    elif 'Error' in l: print(l[:300])
"; done
