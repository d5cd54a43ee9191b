"""Average PMC counters per kernel from a rocprofv3 --pmc counter_collection csv: pmc_summary.py <csv> [kernel substring ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s %16.0f  (x%d)" % (c, sum(v) / len(v), len(v)))
