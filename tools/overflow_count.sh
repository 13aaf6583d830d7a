# how many queries take the overflow kernel: uniform vs clustered synthetic features
for c in 0 0.4 0.7; do timeout -k 10 200 python bench.py --clustered $c --streams 1 --steps 10 --no-cpu --no-e2e --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('clustered $c', 'overflow queries/step', r['overflow_queries_per_step'], 'of', 769*2000, 'pairs/launch', r['scored_pairs_per_launch'], 'fps', round(d['value']))"; done
