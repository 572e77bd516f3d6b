#!/usr/bin/env python3
"""Average per-dispatch PMC counters per kernel from a rocprofv3 --pmc csv output directory."""
import csv, collections, glob, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, "n=%d" % len(next(iter(v.values()))))
    for c, x in sorted(v.items()):
        print("    %-32s %16.0f" % (c, sum(x) / len(x)))
