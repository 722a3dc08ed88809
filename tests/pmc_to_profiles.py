"""Copy the judged numbers of one profiling session (tests/profile_all.sh on the GPU box) from gpurun_out/ into
profiles/: rocprofv3 kernel stats, PMC summaries and the per-launch HBM traffic bench.py reports as roofline.traffic.
usage: python tests/pmc_to_profiles.py <tag e.g. r01_v6> <session dir suffix e.g. 6>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sfx = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles")
for wl in ("atrium", "soup"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof{sfx}", wl, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, f"{tag}_{wl}_kernel_stats.csv"))
    b = os.path.join(ROOT, "gpurun_out", f"prof{sfx}", f"bench_{wl}.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out, f"{tag}_{wl}_bench_under_rocprof.json"))
    s = os.path.join(ROOT, "gpurun_out", f"pmc_{wl}{sfx}", "summary.json")
    if not os.path.exists(s):
        continue
    pmc = json.load(open(s))
    json.dump(pmc, open(os.path.join(out, f"{tag}_{wl}_pmc_summary.json"), "w"), indent=1)
    kname = "k_flat_refill" if "k_flat_refill" in pmc else "k_generation"   # the soup's camera rays run on the lane-refill kernel
    g = pmc[kname]
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests
    # at 64 B, i.e. reports half the bytes of 16 B/lane reads -> doubled.  Separate --pmc passes (tests/pmc_passes.sh).
    traffic = int((2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024)
    json.dump({
        "workload": wl, "kernel": kname, "session": tag,
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tests/pmc_passes.sh), mean over the dispatches of the run",
        "FETCH_SIZE_KiB": g["FETCH_SIZE"], "WRITE_SIZE_KiB": g["WRITE_SIZE"],
        "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane reads (MI355X_MICROARCH.md, HBM section); uncalibrated for this gather pattern",
        "traffic_bytes_per_launch": traffic,
        "TCC_hit_rate": g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"]),
        "L1_hit_rate": 1.0 - g["TCP_TCC_READ_REQ_sum"] / g["TCP_TOTAL_CACHE_ACCESSES_sum"],
        "valu_busy": 4.0 * g["SQ_INSTS_VALU"] / (1024.0 * g["GRBM_GUI_ACTIVE"] / 8.0),
    }, open(os.path.join(out, f"pmc_traffic_{wl}.json"), "w"), indent=1)
    # everything bench.py needs to price the kernel against VALU issue, L1, L2 and HBM limits, per launch
    keys = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT")
    json.dump({
        "workload": wl, "kernel": kname, "session": tag,
        "source": "rocprofv3 --pmc, separate passes (tests/pmc_passes.sh): mean over the dispatches of tests/profile_run.py, node format pinned to the one bench.py reports",
        "units": "FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2, MI355X_MICROARCH.md HBM section); all others are event counts summed over the chip",
        "counters_per_launch": {k: g[k] for k in keys if k in g},
    }, open(os.path.join(out, f"pmc_counters_{wl}.json"), "w"), indent=1)
    print(wl, "traffic", traffic)
