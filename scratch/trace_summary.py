import csv, glob, collections, sys
t = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
prev_end = None; gaps = []
for r in rows:
    k = r["Kernel_Name"][:30]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[k].append(e - s)
    if prev_end is not None: gaps.append(s - prev_end)
    prev_end = e
for k, v in dur.items():
    v2 = sorted(v)
    print(k, len(v), "avg %.1f us  med %.1f  min %.1f  max %.1f" % (sum(v)/len(v)/1e3, v2[len(v2)//2]/1e3, v2[0]/1e3, v2[-1]/1e3))
g = sorted(gaps)
print("gaps: n", len(g), "avg %.1f us med %.1f max %.1f" % (sum(g)/len(g)/1e3, g[len(g)//2]/1e3, g[-1]/1e3))
