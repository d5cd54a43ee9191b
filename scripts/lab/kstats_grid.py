"""Per (kernel, grid) average duration from a rocprofv3 kernel trace, in first-launch order: kstats_grid.py <trace.csv> [substring ...]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg, order = collections.defaultdict(list), []
for r in rows:
    k = r["Kernel_Name"]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    key = (k.split("(")[0][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    if key not in agg:
        order.append(key)
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key in order:
    v = agg[key][1:] or agg[key]
    print("  %8.1f us x %4d  grid %8s wg %4s  %s" % (sum(v) / len(v) / 1e3, len(agg[key]), key[1], key[2], key[0]))
