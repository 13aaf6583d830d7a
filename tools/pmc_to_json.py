"""FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh -> HBM bytes per launch per kernel
(MI355X_MICROARCH.md, HBM section: both counters are in KB; on gfx950 FETCH_SIZE reads exactly half of
wide, 16 B/lane, coalesced reads -> x2; WRITE_SIZE is exact)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha   # noqa: E402  the counters belong to exactly these kernel sources


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    with open(f"{d}/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
main = sys.argv[3] if len(sys.argv) > 3 else "match_union_kernel"
# the command / workload the passes were taken on: the callers that are not tools/profile_round.sh say so (VISO_PMC_COMMAND,
# VISO_PMC_WORKLOAD: tools/image_profile.sh) -- round 4's image file carried the matcher profile's strings
out = {"command": os.environ.get("VISO_PMC_COMMAND") or
                  "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --streams 1 "
                  "--steps 4 --warmup 1 --no-cpu --no-e2e --no-streaming (two separate passes, tools/profile_round.sh)",
       "workload": os.environ.get("VISO_PMC_WORKLOAD") or
                   f"configs[1], {os.environ.get('VISO_PMC_FRAMES', '512')} frame pairs/batch, 2000 kp/image (bench.py defaults)",
       "kernel_source_sha256": kernel_source_sha(),
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reads exactly 1/2 of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE exact",
       "other_kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("match", "sort", "pack", "ransac", "inlier", "circle", "collect", "extract", "harris")):
        continue
    e = {"FETCH_SIZE_KB_per_launch": fetch.get(k, 0.0), "WRITE_SIZE_KB_per_launch": write.get(k, 0.0),
         "hbm_bytes_per_launch_corrected": (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024}
    if main in k:
        out["kernel"] = k
        out.update(e)
    else:
        out["other_kernels"][k] = e
print(json.dumps(out, indent=1))
