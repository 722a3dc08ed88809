"""Copy the judged numbers of one profiling session (tests/profile_all.sh on the GPU box) from gpurun_out/ into
profiles/: rocprofv3 kernel stats, PMC summaries and the per-launch counters bench.py prices its roofline with.
usage: python tests/pmc_to_profiles.py <tag e.g. r03> <session dir suffix e.g. r03a> [workload ...]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sfx = sys.argv[1], sys.argv[2]
wls = sys.argv[3:] or ["atrium", "soup", "cornell", "atrium4k"]
out = os.path.join(ROOT, "profiles")
PINNED = {"atrium": 0, "soup": 5, "cornell": 1}   # RaycaStats.node_format & 5 (bit 0: 4-wide, bit 2: fp16) the session pinned for generation 0
keys = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES",
        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM_RD", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum")
for wl in wls:
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof{sfx}", wl, "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, f"{tag}_{wl}_kernel_stats.csv"))
    if wl == "atrium":
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof{sfx}", "atrium_f4", "*", "*kernel_stats.csv")):
            shutil.copy(f, os.path.join(out, f"{tag}_atrium_inflight_kernel_stats.csv"))
        b = os.path.join(ROOT, "gpurun_out", f"prof{sfx}", "bench_atrium_f4.json")
        if os.path.exists(b):
            shutil.copy(b, os.path.join(out, f"{tag}_atrium_inflight_bench_under_rocprof.json"))
    b = os.path.join(ROOT, "gpurun_out", f"prof{sfx}", f"bench_{wl}.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out, f"{tag}_{wl}_bench_under_rocprof.json"))
    s = os.path.join(ROOT, "gpurun_out", f"pmc_{wl}{sfx}", "summary.json")
    if not os.path.exists(s):
        continue
    pmc = json.load(open(s))
    json.dump(pmc, open(os.path.join(out, f"{tag}_{wl}_pmc_summary.json"), "w"), indent=1)
    kernels = {}
    for kname, g in pmc.items():
        if "FETCH_SIZE" not in g or "SQ_INSTS_VALU" not in g:
            continue
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests
        # at 64 B, i.e. reports half the bytes of 16 B/lane reads -> doubled.  Separate --pmc passes (tests/pmc_passes.sh).
        kernels[kname] = {
            "kernel_name": g.get("kernel_name"),
            "node_format_bits": PINNED.get(wl) if kname in ("k_generation", "k_flat_refill") else None,
            "binary_f32_record_bytes": 48,   # (RaycaStats.node_format bit 12 of the library this session ran: RAYCA_NODE_CH48)
            "launch_ms_under_pmc": round(g["launch_ms_under_pmc"], 4), "dispatches_averaged": g["dispatches_averaged"],
            "hbm_traffic_bytes_per_launch": int((2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024),
            "l1_hit_rate": round(1.0 - g["TCP_TCC_READ_REQ_sum"] / max(g["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0), 4),
            "l2_hit_rate": round(g["TCC_HIT_sum"] / max(g["TCC_HIT_sum"] + g["TCC_MISS_sum"], 1.0), 4),
            "wave_time_waiting": round(g["SQ_WAIT_ANY"] / max(g["SQ_WAVE_CYCLES"], 1.0), 3),
            "valu_lane_activity": round(g["SQ_THREAD_CYCLES_VALU"] / max(64.0 * g["SQ_ACTIVE_INST_VALU"], 1.0), 3),
            "counters_per_launch": {k: g[k] for k in keys if k in g},
        }
    json.dump({
        "workload": wl, "session": tag,
        "source": "rocprofv3 --pmc, separate passes (tests/pmc_passes.sh): mean over the dispatches of tests/profile_run.py (one frame at a time), node format pinned to the one bench.py reports",
        "units": "FETCH_SIZE / WRITE_SIZE in KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2, MI355X_MICROARCH.md HBM section); all others are event counts summed over the chip; "
                 "launch_ms_under_pmc = mean End - Start timestamp of the counted dispatches",
        "kernels": kernels,
    }, open(os.path.join(out, f"pmc_counters_{wl}.json"), "w"), indent=1)
    print(wl, {k: (v["launch_ms_under_pmc"], v["hbm_traffic_bytes_per_launch"]) for k, v in kernels.items()})
