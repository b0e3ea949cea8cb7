import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:36]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if 'rocclr' in k: continue
    print(k, {c: round(v / max(cnt[k][c], 1)) for c, v in agg[k].items()})
t = glob.glob(d + "/*/*kernel_trace.csv")
if t:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        dur[r["Kernel_Name"][:36]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in dur.items():
        if 'rocclr' in k: continue
        print("dur", k, len(v), "avg us", sum(v) / len(v) / 1e3, "max", max(v) / 1e3)
