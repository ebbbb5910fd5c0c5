"""Turn the raw rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/<tag>_*) into the small committed summaries
under profiles/: kernel-stats table, MFMA-busy table, HBM traffic per launch (FETCH_SIZE doubled for 16-byte-per-lane
reads on gfx950, as MI355X_MICROARCH.md prescribes) for the GEMM and the CRF kernels.
usage: python tools/summarize_profiles.py r04"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("pnp::", "").split("(")[0][:80]


CFG_IMAGES = {"psc59": 35, "coco80": 35, "ade768": 8}


def kernel_stats(sub, out, cfg=None):
    rows = list(csv.DictReader(open(os.path.join(G, f"{TAG}_{sub}", "prof_kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats of `python3 bench.py {'--dtype bf16 ' if 'bf16' in sub else ''}"
             f"{'--config ' + cfg + ' ' if cfg else ''}--steps 2 --warmup 1 "
             f"--pipelines 1 --no-cpu-baseline --no-other-modes --no-other-configs --no-noise12 --no-fixture-check` "
             f"({'bf16 throughput mode' if 'bf16' in sub else 'headline mode bf16x3'}; 3 steps of {CFG_IMAGES.get(cfg, 35)} images in the trace)",
             "# kernel | calls | total ms | avg us | min us | max us | % of GPU time"]
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.0005:
            continue
        lines.append(f"{short(r['Name'])} | {r['Calls']} | {t / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                     f"{float(r['MaxNs']) / 1e3:.1f} | {100 * t / tot:.1f}")
    lines.append(f"# total kernel time {tot / 1e6:.2f} ms = {tot / 3e6:.2f} ms per step")
    open(os.path.join(P, out), "w").write("\n".join(lines) + "\n")


def pmc(sub):
    path = os.path.join(G, f"{TAG}_{sub}", "pmc_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    return agg, n


def traffic(fetch_sub, write_sub, note_extra=""):
    fa, fn = pmc(fetch_sub)
    wa, wn = pmc(write_sub)
    tr = {"note": "HBM-side bytes per launch: FETCH_SIZE (KB, doubled: gfx950 reports half of wide coalesced reads) + WRITE_SIZE (KB)"
                  + note_extra, "steps_profiled": 2, "kernels": {}}         # bench.py --steps 1 --warmup 1 under the profiler
    for k in fa:
        c = fn[k]["FETCH_SIZE"]
        w = wa.get(k, {}).get("WRITE_SIZE", 0.0) / max(wn.get(k, {}).get("WRITE_SIZE", 1), 1)
        f = fa[k]["FETCH_SIZE"] / c
        tr["kernels"][k] = {"launches_profiled": c, "fetch_kb_raw": f, "write_kb": w,
                            "traffic_bytes_per_launch": (2 * f + w) * 1024}
    return tr


def main():
    os.makedirs(P, exist_ok=True)
    kernel_stats("trace", f"{TAG}_bench_kernel_stats_summary.txt")
    if os.path.exists(os.path.join(G, f"{TAG}_trace_bf16")):
        kernel_stats("trace_bf16", f"{TAG}_bench_bf16_kernel_stats_summary.txt")
    agg, n = pmc("pmc_mfma")
    out = {"note": "headline mode (bf16x3), per launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles = /8); SQ_VALU_MFMA_BUSY_CYCLES "
                   "is summed over the 1024 SIMDs; mfma_busy_frac = busy / (cycles * 1024): share of SIMD-cycles with the matrix pipe "
                   "busy at the clock the chip actually held", "kernels": {}}
    for k, v in agg.items():
        c = n[k]["GRBM_GUI_ACTIVE"]
        if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / c / 8
        out["kernels"][k] = {"launches": c, "cycles_per_launch": cyc, "mfma_busy_cycles_per_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"] / c,
                             "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / c / (cyc * 1024)}
    json.dump(out, open(os.path.join(P, f"{TAG}_mfma_busy.json"), "w"), indent=1)
    tr = traffic("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", " -- headline mode bf16x3")
    json.dump(tr, open(os.path.join(P, f"{TAG}_hbm_traffic.json"), "w"), indent=1)
    g = {k: v for k, v in tr["kernels"].items() if "gemm_nt_wide" in k or "gemm_nt_x3" in k}
    json.dump({"note": tr["note"], "kernels": g}, open(os.path.join(P, f"{TAG}_gemm_traffic_bf16x3.json"), "w"), indent=1)
    if os.path.exists(os.path.join(G, f"{TAG}_pmc_bf16_FETCH_SIZE")):
        tb = traffic("pmc_bf16_FETCH_SIZE", "pmc_bf16_WRITE_SIZE", " -- bf16 throughput mode")
        json.dump({"note": tb["note"], "kernels": {k: v for k, v in tb["kernels"].items() if "gemm_nt_wide" in k}},
                  open(os.path.join(P, f"{TAG}_gemm_traffic.json"), "w"), indent=1)
    def l2_hit(sub, out):
        try:
            ta, tn = pmc(sub)
        except FileNotFoundError:
            return
        d = {"steps_profiled": 2}
        for k, v in ta.items():
            c = max(tn[k]["TCC_HIT_sum"], 1)
            d[k] = {"l2_hit_rate": v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1), "launches_profiled": c,
                    "tcc_req_per_launch": (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]) / c}
        json.dump(d, open(os.path.join(P, out), "w"), indent=1)

    l2_hit("pmc_tcc", f"{TAG}_crf_l2_hit.json")
    for cfg in CFG_IMAGES:                              # BASELINE configs 3-5 (tools/collect_profiles_configs.sh)
        if os.path.exists(os.path.join(G, f"{TAG}_trace_{cfg}")):
            kernel_stats(f"trace_{cfg}", f"{TAG}_{cfg}_kernel_stats_summary.txt", cfg)
        if os.path.exists(os.path.join(G, f"{TAG}_pmc_{cfg}_FETCH_SIZE")):
            tc = traffic(f"pmc_{cfg}_FETCH_SIZE", f"pmc_{cfg}_WRITE_SIZE", f" -- config {cfg}, headline mode bf16x3, DenseCRF kernels")
            json.dump({"note": tc["note"], "steps_profiled": 2, "kernels": {k: v for k, v in tc["kernels"].items() if "crf_" in k}},
                      open(os.path.join(P, f"{TAG}_{cfg}_crf_traffic.json"), "w"), indent=1)
        l2_hit(f"pmc_{cfg}_tcc", f"{TAG}_{cfg}_crf_l2_hit.json")
    b = os.path.join(G, f"{TAG}_bench.json")
    if os.path.exists(b):
        open(os.path.join(P, f"{TAG}_bench.json"), "w").write(open(b).read())
    print("wrote", sorted(f for f in os.listdir(P) if f.startswith(TAG)))


if __name__ == "__main__":
    main()
