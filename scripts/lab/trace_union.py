"""Union of kernel intervals of a rocprofv3 kernel trace (concurrent streams overlap): busy time vs span over the last fraction of
the trace.  usage: trace_union.py <trace.csv> [fraction=0.5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * (1 - frac)
iv = [x for x in iv if x[0] >= t_lo]
span = max(e for _, e in iv) - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
summed = sum(e - s for s, e in iv)
print("last %.0f %% of the trace: span %.1f ms, at least one kernel running %.1f ms (%.1f %%), summed kernel time %.1f ms (x%.2f overlap)" % (
    100 * frac, span / 1e6, busy / 1e6, 100.0 * busy / span, summed / 1e6, summed / busy))
