"""A/B of environment switches on one box: first-ply search at 32768 / 16384 games and a refilled call of 4 x 32768 games, one process per setting.
   python scratch/ab_env.py "" "AGZ_BIG_LOCK=0" ...     (GAME=gobang|connect4|hex|reversi8, H, T, V)"""
import os, sys, subprocess
for rep in range(int(os.environ.get("REPS", "2"))):
    for setting in sys.argv[1:]:
        env = dict(os.environ)
        for kv in setting.split():
            k, v = kv.split("=")
            env[k] = v
        print(f"[{setting or 'default'}]", end=" ", flush=True)
        subprocess.call([sys.executable, os.path.join(os.path.dirname(__file__), "ab_lib.py"), "--one", "default"], env=env)
