for cfg in "4 3" "8 3" "8 4" "8 6" "8 8"; do set -- $cfg; GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python bench.py --streams $2 --steps 80 --no-cpu --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hwq=$1 streams=$2 matcher fps', round(d['value']), 'e2e fps', round(d['end_to_end']['fps']))"; done
