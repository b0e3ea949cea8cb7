import csv, glob, sys
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n0 = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t0 = int(rows[n0]["Start_Timestamp"])
for r in rows[n0:n0 + 40]:
    print("%-28s q=%s  start %8.1f  end %8.1f  dur %6.1f us  grid %s" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
