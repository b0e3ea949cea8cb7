# per-ply search time of the whole-search kernel for several batch sizes and dispatch overrides (workgroup shape x games per tree wave)
import sys, os, time, subprocess
sizes = [int(x) for x in sys.argv[1].split(",")]
envs = [{}]
for tw in (2, 4):
    for g in (os.environ.get("GPWS", "1,2,4,8").split(",")):
        e = {"AGZ_SMALL_GPW": g}
        e["AGZ_SMALL_MAXL"] = "0" if tw == 4 else "1000000"
        envs.append(e)
for L in sizes:
    row = []
    for e in envs:
        env = dict(os.environ); env.update(e)
        out = subprocess.run([sys.executable, "scratch/prof_search.py", "64", str(L), "4"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        ms = out.split("tree ")[1].split(" ms")[0]
        tag = ("tw%d/g%s" % (4 if e.get("AGZ_SMALL_MAXL") == "0" else 2, e["AGZ_SMALL_GPW"])) if e else "default"
        row.append(f"{tag}: {ms}")
    print(L, " | ".join(row), flush=True)
