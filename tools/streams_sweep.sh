for s in 1 2 3 4 6; do timeout -k 10 200 python bench.py --streams $s --steps 60 --no-cpu --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams=$s matcher fps', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'e2e fps', round(d['end_to_end']['fps']))"; done
