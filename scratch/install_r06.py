#!/usr/bin/env python3
"""gpurun_out/<dir> (scratch/collect_r06.sh) -> profiles/<prefix>_*: the judged copies, and the "generation" entries of profiles/pmc_traffic.json
(what bench.py reads for roofline.traffic / roofline_valu).   usage: install_r06.py gpurun_out/r06 r06"""
import glob, json, os, re, shutil, sys
src, pre = sys.argv[1], sys.argv[2]
P = "profiles"
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, f"{pre}_{os.path.basename(f)}"))
for f in ("phase_cycles_stamps", "util_cfg0", "util_cfg2", "util_cfg3"):
    if os.path.exists(os.path.join(src, f + ".txt")):
        name = {"util_cfg0": "pmc_unit_utilisation_128x6", "util_cfg2": "pmc_unit_utilisation_connect4", "util_cfg3": "pmc_unit_utilisation_512x8"}.get(f, f)
        shutil.copy(os.path.join(src, f + ".txt"), os.path.join(P, f"{pre}_{name}.txt"))
for d in sorted(glob.glob(os.path.join(src, "stats_*"))):
    st = glob.glob(os.path.join(d, "*kernel_stats.csv")) + glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(P, f"{pre}_kernel_stats_bench_{os.path.basename(d)[6:]}.csv"))
shutil.copy(os.path.join(src, "pmc_refill_summary.txt"), os.path.join(P, f"{pre}_pmc_refilled_call_by_variant.txt"))
txt = open(os.path.join(src, "pmc_refill_summary.txt")).read()
keys = {0: "gobang9x9_nvict5_64_128x6", 2: "connect4_64_128x6", 3: "gobang9x9_nvict5_64_512x8", 4: "hex9x9_128_512x8", 5: "reversi8_64_512x8"}
pm = json.load(open(os.path.join(P, "pmc_traffic.json")))
blocks = re.split(r"# cfg (\d): rocprofv3 --pmc ([A-Z_0-9a-z ]+?) --kernel-trace[^\n]*\n", txt)[1:]
acc = {}
for i in range(0, len(blocks), 3):
    cfg, ctrs, body = int(blocks[i]), blocks[i + 1].split(), blocks[i + 2]
    e = acc.setdefault(cfg, {"launches": 0})
    for m in re.finditer(r"sum (k_(?:search|selfplay)\w+)<[^>]*> launches (\d+) (\{[^}]*\})", body):
        d = eval(m.group(3))
        for c, v in d.items(): e[c] = e.get(c, 0) + v
        if ctrs[0] == "FETCH_SIZE": e["launches"] += int(m.group(2)); e["kernel"] = m.group(1)
    mj = re.search(r"(\{\"cfg\".*\})", body)
    if mj:
        j = json.loads(mj.group(1))
        e["alg"], e["rollouts"], e["gens"] = j["algorithmic_bytes_of_the_call"], j["rollouts"], j["gens"]
for cfg, e in acc.items():
    if "FETCH_SIZE" not in e or "WRITE_SIZE" not in e or "alg" not in e:
        print("incomplete passes for cfg", cfg, {k: v for k, v in e.items() if k != "kernel"}); continue
    f, w = e["FETCH_SIZE"], e["WRITE_SIZE"]
    pm.setdefault(keys[cfg], {})["generation"] = {
        "source": f"profiles/{pre}_pmc_refilled_call_by_variant.txt",
        "what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* (one pass each, --kernel-trace only) over ONE agz_selfplay call of %d x 32768 games on 32768 "
                "slots (finished games' slots refilled: bench.py's scheduling — the persistent self-play kernel; scratch/pmc_refill.py CFG=%d), summed over the %d launch(es) of %s; algorithmic "
                "bytes of the same call from the device counters" % (e["gens"], cfg, e["launches"], e["kernel"]),
        "fetch_size_kb_raw": f, "write_size_kb": w, "algorithmic_bytes": e["alg"], "rollouts": e["rollouts"], "search_launches": e["launches"],
        "fetch_correction": "x2 (gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane reads; an upper bound here: part of the reads are narrower)",
        "traffic_over_algorithmic": (2 * f + w) * 1024 / e["alg"], "traffic_over_algorithmic_uncorrected": (f + w) * 1024 / e["alg"],
        "valu_insts_per_rollout": e["SQ_INSTS_VALU"] / e["rollouts"], "mfma_insts_per_rollout": e["SQ_INSTS_MFMA"] / e["rollouts"],
        "vmem_rd_insts_per_rollout": e["SQ_INSTS_VMEM_RD"] / e["rollouts"], "lds_insts_per_rollout": e["SQ_INSTS_LDS"] / e["rollouts"],
        "l2_hit_rate": (e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1)) if "TCC_HIT_sum" in e else None}
    print(keys[cfg], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pm[keys[cfg]]["generation"].items() if k not in ("what", "fetch_correction", "source")})
json.dump(pm, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
