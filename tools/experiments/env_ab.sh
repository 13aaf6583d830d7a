#!/bin/bash
# tools/experiments/env_ab.sh "VAR=a" "VAR=b" [reps]: the per-call loop (tools/dropin_probe.py) alternating between two settings of an
# environment knob of the product library, on one box
cd $GRAFT_REPO_ROOT
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for e in "$A" "$B"; do
    printf "%s: " "$e"; env $e python3 tools/dropin_probe.py 257 2000 2>&1 | grep "^frames" | cut -c1-70
  done
done
