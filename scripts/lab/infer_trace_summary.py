"""Summarise a rocprofv3 kernel trace of scripts/lab/infer_busy.py <n> <scale | all>: GPU-busy vs span per image over the last n
images (image boundaries = every `per`-th aff_refine launch: one per scale-pass, i.e. 1 for a single scale, 4 for `all`).
usage: infer_trace_summary.py <trace.csv> [n] [top] [per] [images per unit]   (a unit = one call's worth of passes: an image, or --
with the list walk's default batches, scripts/lab/infer_list_busy.py -- a batch of 8; figures are printed per IMAGE)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("aff_refine_kernel")]
per = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ipu = int(sys.argv[5]) if len(sys.argv) > 5 else 1
first = marks[-per * n - 1] + 1
n_units, n = n, n * ipu
last = rows[first:marks[-1] + 1]
t0, t1 = int(last[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in last)
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(dur(r) for r in last)
print("%d kernels/image; span %.2f ms/image, GPU busy %.2f ms/image (%.0f %%)" % (len(last) / n, (t1 - t0) / n / 1e6, busy / n / 1e6,
                                                                                 100.0 * busy / (t1 - t0)))
agg, cnt = collections.Counter(), collections.Counter()
for r in last:
    k = r["Kernel_Name"][:64] + " wg" + str(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    agg[k] += dur(r)
    cnt[k] += 1
for k, v in agg.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print("  %8.3f ms %6.1f x %7.1f us  %s" % (v / n / 1e6, cnt[k] / n, v / cnt[k] / 1e3, k))
