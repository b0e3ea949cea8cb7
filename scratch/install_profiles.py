#!/usr/bin/env python3
"""gpurun_out/<dir> (scratch/collect_profiles.sh) -> profiles/<prefix>_*: the judged copies, and profiles/pmc_traffic.json (what bench.py
reads for roofline.traffic / roofline_valu) from the PMC passes.  usage: install_profiles.py gpurun_out/r03a r03"""
import glob, json, os, re, shutil, sys
src, pre = sys.argv[1], sys.argv[2]
P = "profiles"
def cp(a, b):
    if os.path.exists(os.path.join(src, a)): shutil.copy(os.path.join(src, a), os.path.join(P, f"{pre}_{b}"))
cp("bench_headline.json", "bench_headline.json")
for c in (2, 3, 4, 5): cp(f"bench_config{c}.json", f"bench_config{c}.json")
cp("bench_under_rocprof.json", "bench_under_rocprof.json")
cp("per_ply_by_batch.txt", "per_ply_by_batch.txt")
cp("pmc_summary.txt", "pmc_first_ply_search.txt")
st = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")) + glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if st: shutil.copy(st[0], os.path.join(P, f"{pre}_kernel_stats_full_generation.csv"))
# PMC -> pmc_traffic.json
txt = open(os.path.join(src, "pmc_summary.txt")).read()
def counter(name):
    m = re.search(r"k_search_small[^\n]*'" + name + r"': (\d+)", txt)
    return int(m.group(1)) if m else None
m = re.search(r'"sum_p": (\d+), "sum_new": (\d+), "rollouts": (\d+).*"algorithmic_bytes_per_search_launch": ([0-9.e+]+)', txt)
alg, rollouts = float(m.group(4)), int(m.group(3))
fetch, write, valu = counter("FETCH_SIZE"), counter("WRITE_SIZE"), counter("SQ_INSTS_VALU")
hit, miss = counter("TCC_HIT_sum"), counter("TCC_MISS_sum")
key = "gobang9x9_nvict5_64_128x6"
d = {key: {
    "what": "rocprofv3 --pmc passes (separate runs, each with --kernel-trace only) of k_search_small<0,2,12,128,4,4>: one launch = one mcts_single at L=32768, V=64, Gobang 9x9, 128x6 (scratch/pmc_point.py, 2 launches averaged); see %s_pmc_first_ply_search.txt" % pre,
    "fetch_size_kb_per_launch_raw": fetch, "write_size_kb_per_launch": write,
    "fetch_correction": "x2: gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section). The item rows are read with global_load_dwordx4/x3, rank / child bytes, edge entries and next words with narrower loads, so the corrected figure is an upper bound and the raw one a lower bound.",
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024, "traffic_bytes_per_launch_uncorrected": (fetch + write) * 1024,
    "algorithmic_bytes_per_launch_at_profile_point": alg,
    "traffic_over_algorithmic": (2 * fetch + write) * 1024 / alg, "traffic_over_algorithmic_uncorrected": (fetch + write) * 1024 / alg,
    "tcc_hit_rate": (hit / (hit + miss)) if hit and miss else None,
    "valu_insts_per_launch": valu, "valu_insts_per_rollout": valu / rollouts if valu else None}}
# BASELINE configs 2..5 (scratch/pmc_configs.sh): tree traffic over algorithmic bytes, HBM bytes per leaf of the wide-trunk network
cs = os.path.join(src, "cfg_summary.txt")
if os.path.exists(cs):
    shutil.copy(cs, os.path.join(P, f"{pre}_pmc_configs_first_ply.txt"))
    ctxt = open(cs).read()
    keys = {2: "connect4_64_128x6", 3: "gobang9x9_nvict5_64_512x8", 4: "hex9x9_128_512x8", 5: "reversi8_64_512x8"}
    blocks = re.split(r"# cfg (\d) (FETCH_SIZE|WRITE_SIZE)\n", ctxt)[1:]
    data = {}
    for i in range(0, len(blocks), 3):
        cfg, ctr, body = int(blocks[i]), blocks[i + 1], blocks[i + 2]
        e = data.setdefault(cfg, {})
        for m in re.finditer(r"sum void agz::(\w+) (\d+) launches (\d+)", body):
            e[(m.group(1), ctr)] = int(m.group(2))
        m = re.search(r'"algorithmic_bytes_per_search_launch": ([0-9.e+]+), "nn_leaves": (\d+)', body)
        e["alg"], e["leaves"] = float(m.group(1)), int(m.group(2))
    for cfg, e in data.items():
        tree = [k for k in ("k_search_small", "k_search_big", "k_rollout_eager") if (k, "FETCH_SIZE") in e][0]
        f, w = e[(tree, "FETCH_SIZE")] / 2, e[(tree, "WRITE_SIZE")] / 2          # two searches per pass
        o = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one first-ply search at 32768 games (scratch/pmc_configs.sh, CFG=%d; the whole-search kernel's launch: tree step and network together for the wide trunks); sums over the launches of the search; see %s_pmc_configs_first_ply.txt" % (cfg, pre),
             "tree_kernel": tree, "fetch_size_kb_per_search_raw": f, "write_size_kb_per_search": w, "algorithmic_bytes_per_search": e["alg"],
             "traffic_over_algorithmic": (2 * f + w) * 1024 / e["alg"], "traffic_over_algorithmic_uncorrected": (f + w) * 1024 / e["alg"]}
        if ("k_mlp_big", "FETCH_SIZE") in e:
            nf, nw = e[("k_mlp_big", "FETCH_SIZE")], e[("k_mlp_big", "WRITE_SIZE")]
            o["nn_hbm_bytes_per_leaf"] = (2 * nf + nw) * 1024 / (2 * e["leaves"])
            o["nn_hbm_bytes_per_leaf_uncorrected"] = (nf + nw) * 1024 / (2 * e["leaves"])
        d[keys[cfg]] = o
json.dump(d, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(d[key], indent=1))
