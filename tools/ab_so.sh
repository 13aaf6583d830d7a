#!/bin/bash
# tools/ab_so.sh A.so B.so ... : matcher-only bench lines of alternative builds of the library, one after the other in one call
for so in "$@"; do
  VISO_HIP_SO=$so python bench.py --steps 20 --warmup 5 --no-cpu --no-streaming --no-images --no-e2e 2>/dev/null > gpurun_out/ab_tmp.json || true
  python - "$so" <<EOP
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"]), d["ms_per_step"], d["roofline"].get("kernel_ms"))
EOP
done
