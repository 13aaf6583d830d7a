for f in 0 1 5 7 15 31; do VISO_STRIP_DEBUG=$f timeout -k 10 200 python bench.py --matcher 4 --streams 1 --steps 20 --no-cpu --no-streaming --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$f', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4))"; done
