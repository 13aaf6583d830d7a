#!/bin/bash
# tools/e2e_ab.sh "ENV1=.. ENV2=.." ... : one short bench per environment string (matcher-only + end-to-end), in ONE gpurun call
for envs in "$@"; do
  env $envs python bench.py --steps 40 --warmup 5 --no-cpu --no-streaming --no-images $EXTRA 2>/dev/null > gpurun_out/ab_tmp.json || true
  python - "$envs" <<EOP
import json,sys
try:
    d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().split("\n")[-1])
    print(sys.argv[1], "| matcher", round(d["value"]), "e2e", round(d["end_to_end"]["fps"]), "ms", round(d["end_to_end"]["ms_per_step"],3), "kernel_ms", round(d["roofline"]["kernel_ms"],3), flush=True)
except Exception as e:
    print(sys.argv[1], "| failed", e, flush=True)
EOP
done
