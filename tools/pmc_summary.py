"""Average the rocprofv3 counter_collection.csv files written by tools/pmc_batch.sh per kernel."""
import collections
import csv
import json
import sys

out = {}
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f"{d}/p_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
keep = {k: v for k, v in out.items() if "match" in k or "ransac" in k or "pack" in k or "sort" in k}
print(json.dumps(keep, indent=1))
