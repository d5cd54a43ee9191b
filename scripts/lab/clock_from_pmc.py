"""Effective shader clock per kernel from a rocprofv3 `--pmc GRBM_GUI_ACTIVE --kernel-trace` run of bench.py:
clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md "DVFS give-back"; reads high on dispatches shorter than
~0.3 ms, so only longer ones are averaged).  usage: clock_from_pmc.py <dir with *_counter_collection.csv and *_kernel_trace.csv> [min_ms]"""
import collections, csv, glob, sys
d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    ns, name = dur.get(r["Dispatch_Id"], (0, ""))
    if ns >= min_ms * 1e6:
        agg[name[:70]].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
print("effective clock (GHz) = GRBM_GUI_ACTIVE / 8 / duration, dispatches >= %.2f ms" % min_ms)
tot_c, tot_n = 0.0, 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    cl = [x[0] for x in v]
    print("  %-70s n=%4d  avg %.3f  min %.3f  max %.3f  (avg %.3f ms)" % (k, len(v), sum(cl) / len(cl), min(cl), max(cl), sum(x[1] for x in v) / len(v) / 1e6))
    tot_c += sum(x[0] * x[1] for x in v); tot_n += sum(x[1] for x in v)
if tot_n:
    print("time-weighted over all listed dispatches: %.3f GHz" % (tot_c / tot_n))
