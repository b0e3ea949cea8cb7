# summary of a rocprofv3 --pmc / --kernel-trace output directory written with -o x --output-format csv (files directly in the directory)
import csv, collections, sys, glob
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:44]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    if 'rocclr' in k: continue
    print(k, {c: round(v / max(cnt[k][c], 1)) for c, v in agg[k].items()})
t = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
if t:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t[0])):
        dur[r["Kernel_Name"][:44]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in dur.items():
        if 'rocclr' in k: continue
        print("dur", k, len(v), "avg us", sum(v) / len(v) / 1e3, "max", max(v) / 1e3)
