"""Per kernel, from ONE rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` over bench.py: the effective
shader clock (GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration), the matrix-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8)) and their product against the 2.4 GHz peak -- what a kernel can reach at the clock it is given.
usage: mfma_busy_from_pmc.py <dir> [min_ms]"""
import collections, csv, glob, sys
d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for did, c in vals.items():
    ns, name = dur.get(did, (0, ""))
    if ns >= min_ms * 1e6 and "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        agg[name[:64]].append((cyc / ns, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), ns))
print("dispatches >= %.2f ms: clock GHz = GRBM_GUI_ACTIVE / 8 / duration; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)" % min_ms)
print("%-64s %5s %8s %7s %7s %s" % ("kernel", "n", "avg ms", "GHz", "busy", "clock/2.4 x busy"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(x[2] for x in kv[1])):
    n = len(v)
    ghz = sum(x[0] for x in v) / n
    busy = sum(x[1] for x in v) / n
    print("%-64s %5d %8.3f %7.3f %7.3f %7.3f" % (k, n, sum(x[2] for x in v) / n / 1e6, ghz, busy, ghz / 2.4 * busy))
