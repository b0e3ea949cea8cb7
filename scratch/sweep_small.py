# per-ply search time of the whole-search kernel for several batch sizes and dispatch overrides
import sys, os, time, subprocess
sizes = [int(x) for x in sys.argv[1].split(",")]
envs = [{}] + [{"AGZ_SMALL_GPW": str(k)} for k in (os.environ.get("GPWS", "1,2,4,8").split(","))] + [{"AGZ_SMALL_MAXL": "0"}]
for L in sizes:
    row = []
    for e in envs:
        env = dict(os.environ); env.update(e)
        out = subprocess.run([sys.executable, "scratch/prof_search.py", "64", str(L), "3"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        ms = out.split("wall ")[1].split(" ms")[0]
        row.append(f"{'/'.join(k[10:]+'='+v for k, v in e.items()) or 'default'}: {ms}")
    print(L, " | ".join(row), flush=True)
