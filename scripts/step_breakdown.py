"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py: uses the consistency kernel (one launch per step) as
the step marker and averages over the last few steps, so first-run library searches do not pollute the numbers."""
import csv, sys, collections, re
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [int(r['Start_Timestamp']) for r in rows if 'consistency_fwd' in r['Kernel_Name'] or 'cons_fwd' in r['Kernel_Name']]
# keep only markers that open a real training step (>= 10 ms to the next marker); the roofline probe launches the same
# kernel back to back at the end of the run
real = [i for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 10e6]
assert len(real) > nsteps + 1, len(real)
t0, t1 = marks[real[-1 - nsteps]], marks[real[-1]]          # the last interval may run into the probe: leave it out
d = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r['Start_Timestamp'])
    if t0 <= s < t1:
        n = re.sub(r'\(.*', '', r['Kernel_Name'])[:80]
        d[n][0] += 1; d[n][1] += (int(r['End_Timestamp']) - s) / 1e6
tot = sum(v[1] for v in d.values())
print("wall %.2f ms/step, kernel busy %.2f ms/step" % ((t1 - t0) / 1e6 / nsteps, tot / nsteps))
for n, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-80s %5d %7.3f ms/step" % (n, v[0] // nsteps, v[1] / nsteps))
