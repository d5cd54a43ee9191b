"""Per-kernel average duration from a rocprofv3 kernel trace: kstats.py <trace.csv> [substring ...] (skips each kernel's first 3 launches)"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen, agg = collections.Counter(), collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    seen[k] += 1
    if seen[k] > 3:
        agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %8.1f us x %4d  %s" % (sum(v) / len(v) / 1e3, len(v), k[:90]))
