"""Per-step breakdown by (kernel, grid): like step_breakdown.py, but launches of one kernel with different grids are listed apart
and with their workgroup counts -- small grids with long durations are the launches that leave the chip idle.
usage: step_breakdown_grid.py <kernel_trace.csv> [nsteps] [min_us_per_step]"""
import csv, sys, collections, re
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [int(r['Start_Timestamp']) for r in rows if 'cons_fwd' in r['Kernel_Name'] or 'consistency_fwd' in r['Kernel_Name']]
real = [i for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 10e6]
assert len(real) > nsteps + 1, len(real)
t0, t1 = marks[real[-1 - nsteps]], marks[real[-1]]
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r['Start_Timestamp'])
    if t0 <= s < t1:
        n = re.sub(r'\(.*', '', r['Kernel_Name'])[:56]
        gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        wx = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        k = (n, gx // max(wx, 1), wx)
        d[k][0] += 1; d[k][1] += (int(r['End_Timestamp']) - s) / 1e3
print("wall %.2f ms/step" % ((t1 - t0) / 1e6 / nsteps))
print("%-56s %8s %5s %6s %9s %10s" % ("kernel", "wgs", "thr", "n/step", "avg us", "us/step"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    if v[1] / nsteps < min_us:
        continue
    print("%-56s %8d %5d %6.1f %9.1f %10.1f" % (k[0], k[1], k[2], v[0] / nsteps, v[1] / v[0], v[1] / nsteps))
