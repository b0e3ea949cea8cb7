# BASELINE.md section 3 rows from the committed bench lines (profiles/<round>_bench_*.json).  usage: baseline_table.py r04
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1] if len(sys.argv) > 1 else "r04"
def load(n):
    return json.loads(open(os.path.join(ROOT, "profiles", n)).read().strip().splitlines()[-1])
rows = [("Gobang 9×9/5, 32768×64, 128×6 (**headline**)", f"{pre}_bench_headline.json"), ("Connect4, 32768×64, 128×6", f"{pre}_bench_config2.json"),
        ("Gobang 9×9/5, 32768×64, 512×8", f"{pre}_bench_config3.json"), ("Hex 9×9, 32768×128, 512×8", f"{pre}_bench_config4.json"),
        ("Reversi 8×8, 32768×64, 512×8", f"{pre}_bench_config5.json")]
print("| Config | GPUs | rollouts/s (`value`: refilled calls) | lock-step generations | search kernels only | with samples in the host `PoolSample` | HBM frac (tree) | MFMA frac (net) | HBM traffic / algorithmic (PMC, whole call) | CPU baseline rollouts/s (threads) |")
print("|---|---|---|---|---|---|---|---|---|---|")
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
keys = ["gobang9x9_nvict5_64_128x6", "connect4_64_128x6", "gobang9x9_nvict5_64_512x8", "hex9x9_128_512x8", "reversi8_64_512x8"]
for (name, f), key in zip(rows, keys):
    d = load(f)
    r, o = d["roofline"], d["roofline_other"]
    hbm = r if r["bound"] == "hbm" else o
    mf = r if r["bound"] == "mfma" else o
    cb = d.get("cpu_baseline") or {}
    g = pm.get(key, {}).get("generation", {})
    print(f"| {name} | 1 | {d['value']/1e6:.1f} M ({d['ms_per_step']:.0f} ms / generation) | {(d.get('value_lockstep_generations') or 0)/1e6:.1f} M | "
          f"{d['rank0']['search_only_rollouts_per_s']/1e6:.1f} M | {(d.get('value_with_host_delivery') or 0)/1e6:.1f} M | {hbm['frac']:.3f} | {mf['frac']:.3f} | "
          f"{g.get('traffic_over_algorithmic_uncorrected', 0):.2f} – {g.get('traffic_over_algorithmic', 0):.2f} | {cb.get('value', 0)/1e3:.1f} K ({cb.get('cores','-')}) |")
