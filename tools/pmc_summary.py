"""Average the rocprofv3 counter_collection.csv files written by tools/pmc_batch.sh per kernel."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha   # noqa: E402  the counters belong to exactly these kernel sources

out = {}
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f"{d}/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
keep = {k: v for k, v in out.items() if "match" in k or "ransac" in k or "pack" in k or "sort" in k}
print(json.dumps({"kernel_source_sha256": kernel_source_sha(),
                  "workload": f"configs[1], {os.environ.get('VISO_PMC_FRAMES', '512')} frame pairs/batch, 2000 kp/image, one stream "
                              "(tools/pmc_batch.sh)",
                  "kernels": keep}, indent=1))
