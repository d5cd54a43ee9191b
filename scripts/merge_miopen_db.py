"""Merge a recorded MIOpen user db (scripts/record_miopen_db.sh) into the shipped one: lines are `key=value`, one per
convolution problem; recorded entries replace shipped ones with the same key.  usage: merge_miopen_db.py <recorded dir>"""
import os, sys
src = sys.argv[1]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "acr_wsss_amd", "miopen_db")
for f in sorted(os.listdir(src)):
    if not f.endswith(".txt"):
        continue
    merged = {}
    for path in (os.path.join(dst, f), os.path.join(src, f)):
        if os.path.exists(path):
            for ln in open(path):
                ln = ln.rstrip("\n")
                if "=" in ln:
                    k, v = ln.split("=", 1)
                    merged[k] = v
    with open(os.path.join(dst, f), "w") as out:
        for k in sorted(merged):
            out.write("%s=%s\n" % (k, merged[k]))
    print(f, len(merged), "entries")
