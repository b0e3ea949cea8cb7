#!/usr/bin/env python3
"""gpurun_out/<dir> (scratch/collect_r05.sh) -> profiles/<prefix>_*: the judged copies, and the "generation" entries of profiles/pmc_traffic.json
(what bench.py reads for roofline.traffic / roofline_valu): separate FETCH_SIZE / WRITE_SIZE / instruction-mix passes over ONE refilled
agz_selfplay call of 2 x 32768 games per configuration, summed over every search launch of the call, against the algorithmic bytes of the same
call from the device counters.   usage: install_r05.py gpurun_out/r05p r05"""
import glob, json, os, re, shutil, sys
src, pre = sys.argv[1], sys.argv[2]
P = "profiles"
for f in ("bench_headline", "bench_headline_noage", "bench_headline_plyloop", "bench_headline_lockstep", "bench_headline_exchange_1rank", "bench_config2", "bench_config3", "bench_config4", "bench_config5", "bench_under_rocprof",
          "bench_config3_two_workgroups", "bench_config4_two_workgroups", "bench_config5_two_workgroups", "bench_headline_tw8", "bench_config3_lockstep"):
    if os.path.exists(os.path.join(src, f + ".json")) and os.path.getsize(os.path.join(src, f + ".json")) > 0:
        shutil.copy(os.path.join(src, f + ".json"), os.path.join(P, f"{pre}_{f}.json"))
for f in ("workgroup_spread", "persistent_phase_shares"):
    if os.path.exists(os.path.join(src, f + ".txt")):
        shutil.copy(os.path.join(src, f + ".txt"), os.path.join(P, f"{pre}_{f}.txt"))
st = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv")) + glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if st: shutil.copy(st[0], os.path.join(P, f"{pre}_kernel_stats_bench_headline.csv"))
shutil.copy(os.path.join(src, "pmc_refill_summary.txt"), os.path.join(P, f"{pre}_pmc_refilled_call_by_variant.txt"))
txt = open(os.path.join(src, "pmc_refill_summary.txt")).read()
keys = {0: "gobang9x9_nvict5_64_128x6", 2: "connect4_64_128x6", 3: "gobang9x9_nvict5_64_512x8", 4: "hex9x9_128_512x8", 5: "reversi8_64_512x8"}
pm = json.load(open(os.path.join(P, "pmc_traffic.json")))
blocks = re.split(r"# cfg (\d): rocprofv3 --pmc ([A-Z_0-9 ]+?) --kernel-trace[^\n]*\n", txt)[1:]
acc = {}
for i in range(0, len(blocks), 3):
    cfg, ctrs, body = int(blocks[i]), blocks[i + 1].split(), blocks[i + 2]
    e = acc.setdefault(cfg, {"launches": 0})
    for m in re.finditer(r"sum (k_(?:search|selfplay)\w+)<[^>]*> launches (\d+) (\{[^}]*\})", body):
        d = eval(m.group(3))
        for c, v in d.items(): e[c] = e.get(c, 0) + v
        if ctrs[0] == "FETCH_SIZE": e["launches"] += int(m.group(2)); e["kernel"] = m.group(1)
    j = json.loads(re.search(r"(\{\"cfg\".*\})", body).group(1))
    e["alg"], e["rollouts"], e["gens"] = j["algorithmic_bytes_of_the_call"], j["rollouts"], j["gens"]
for cfg, e in acc.items():
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
