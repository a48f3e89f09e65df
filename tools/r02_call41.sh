for cfg in "256 8" "256 16" "512 4" "512 8"; do set -- $cfg; LSDR_VIT_WO=$2 LSDR_C3_BATCH_MSAMPLES=$1 timeout 100 python tools/more_one.py c3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('c3 '):
        d = json.loads(l[3:]); print('$cfg', d['value'], d['seconds'], d['host_seconds_per_stage'], d['viterbi'], d['ts_check']['pass'])
    elif 'Error' in l or 'error' in l: print(l[:300])
"; done
