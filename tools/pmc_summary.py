#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc CSV (counter_collection.csv)."""
import csv, re, sys
from collections import defaultdict
path, counter = sys.argv[1], sys.argv[2]
agg = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
            agg[name].append(float(r["Counter_Value"]))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:72s} launches {len(v):4d}  avg {sum(v)/len(v):16.1f}  last {v[-1]:16.1f}")
