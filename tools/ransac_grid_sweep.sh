#!/bin/bash
# tools/ransac_grid_sweep.sh "a,b" "c,d" ...: end-to-end bench line for each VISO_RANSAC_GRID=inlier_blocks,refit_blocks
# (0 = uncapped), alternating in ONE gpurun call.  EXTRA="..." adds bench flags.
for g in "$@"; do
  VISO_RANSAC_GRID=$g python bench.py --steps 40 --warmup 5 --no-cpu --no-streaming --no-images $EXTRA 2>/dev/null > gpurun_out/sweep_tmp.json || true
  python - "$g" <<EOP
import json,sys
d=json.loads(open("gpurun_out/sweep_tmp.json").read().strip().split("\n")[-1])
print("grid", sys.argv[1], "matcher", round(d["value"]), "e2e", round(d["end_to_end"]["fps"]), "ms", round(d["end_to_end"]["ms_per_step"],3), flush=True)
EOP
done
