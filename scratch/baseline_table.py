# BASELINE.md section 3 rows from the committed bench lines (profiles/r03_bench_*.json)
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def load(n):
    return json.loads(open(os.path.join(ROOT, "profiles", n)).read().strip().splitlines()[-1])
rows = [("Gobang 9×9/5, 32768×64, 128×6 (**headline**)", "r03_bench_headline.json"), ("Connect4, 32768×64, 128×6", "r03_bench_config2.json"),
        ("Gobang 9×9/5, 32768×64, 512×8", "r03_bench_config3.json"), ("Hex 9×9, 32768×128, 512×8", "r03_bench_config4.json"),
        ("Reversi 8×8, 32768×64, 512×8", "r03_bench_config5.json")]
print("| Config | GPUs | rollouts/s (generation = `value`) | rollouts/s (search kernels only) | rollouts/s (+ samples delivered to host) | HBM frac (tree) | MFMA frac (net) | CPU baseline rollouts/s (threads) | parity |")
print("|---|---|---|---|---|---|---|---|---|")
for name, f in rows:
    d = load(f)
    r, o = d["roofline"], d["roofline_other"]
    hbm = r if r["bound"] == "hbm" else o
    mf = r if r["bound"] == "mfma" else o
    hd = d["rank0"].get("host_delivery") or {}
    cb = d.get("cpu_baseline") or {}
    print(f"| {name} | 1 | {d['value']/1e6:.1f} M ({d['ms_per_step']:.0f} ms) | {d['rank0']['search_only_rollouts_per_s']/1e6:.1f} M | "
          f"{hd.get('rollouts_per_s_with_host_delivery', 0)/1e6:.1f} M | {hbm['frac']:.3f} | {mf['frac']:.3f} | "
          f"{cb.get('value', 0)/1e3:.1f} K ({cb.get('cores','-')}) | bit-exact vs oracle (bf16 mode incl. MFMA model) |")
