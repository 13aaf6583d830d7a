#!/bin/bash
# tools/experiments/trace_ab.sh "ENV=a" "ENV=b" ... : the stereo call's host-side phases ($VISO_PLAIN_TRACE=1, averaged over the probe's 257
# frames: steadier than the loop's frames/s) for each setting of an environment knob, twice, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for e in "$@"; do
    env $e VISO_PLAIN_TRACE=1 python3 tools/dropin_probe.py 257 2000 2>&1 | python3 -c "
import sys,re
txt=sys.stdin.read()
loop=txt.split('(the loop:)')[1]
m=re.search(r'2 image\(s\) uploaded, (\d+) calls:(.*)\(us per call\)',loop)
v=[float(x) for x in re.findall(r'([0-9.]+)(?=  |\s*$)',m.group(2).replace('  ',' ; '))] if False else [float(x) for x in re.findall(r' ([0-9]+\.[0-9])',m.group(2))]
fps=re.search(r'fps ([0-9.]+)',txt).group(1)
print('$e: stereo call %.1f us = %s ; loop %.0f frames/s' % (sum(v), ' + '.join('%.1f'%x for x in v), float(fps)))
"
  done
done
