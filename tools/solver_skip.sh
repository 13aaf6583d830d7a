for f in 0 2 3 7 15; do VISO_SOLVER_SKIP=$f timeout -k 10 200 python bench.py --steps 60 --no-cpu --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skip=$f e2e fps', round(d['end_to_end']['fps']), 'matcher', round(d['value']))"; done
