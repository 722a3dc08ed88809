"""Summary of one tests/microbench/run_peaks.sh session (gpurun_out/peaks_<tag>/) -> profiles/peaks_<round>.json, the file
bench.py takes the VALU-issue and L1 ceilings from, plus the rocprofv3 files it is derived from.
usage: python tests/microbench/peaks_to_profiles.py <tag e.g. r03b> [round e.g. r03]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"peaks_{tag}")
out = os.path.join(ROOT, "profiles")
rows = [json.loads(l) for l in open(os.path.join(src, "peaks.jsonl")) if l.startswith("{")]
valu = collections.OrderedDict()
for r in rows:
    if r["bench"] == "valu":
        valu.setdefault(r["op"], {})[str(r["waves_per_simd"])] = r["simd_cycles_per_instr_at_max_clock"]
mix = next(v for k, v in valu.items() if k.startswith("slab mix"))
l1 = [r for r in rows if r["bench"] == "l1"]


def dispatches(sub):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])) if f else ():
        d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(disp.values())


# PMC: what one TCP_TOTAL_CACHE_ACCESSES is worth, and how many the L1 retires per clock and CU
l1_pmc = collections.OrderedDict()
for d in dispatches("pmc_l1"):
    if "k_l1<" not in d["kernel"] or not d.get("SQ_INSTS_VMEM_RD"):
        continue
    mode = int(d["kernel"].split("k_l1<")[1].split(">")[0])
    w = d["grid"] // (256 * 256)
    key = f"mode{mode}_w{w}"
    if key in l1_pmc:
        continue
    clk = d["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
    l1_pmc[key] = {"load_instr": int(d["SQ_INSTS_VMEM_RD"]), "cache_accesses_per_load_instr": round(d["TCP_TOTAL_CACHE_ACCESSES_sum"] / d["SQ_INSTS_VMEM_RD"], 2),
                   "total_accesses_per_load_instr": round(d["TCP_TOTAL_ACCESSES_sum"] / d["SQ_INSTS_VMEM_RD"], 2),
                   "l2_reads_per_load_instr": round(d["TCP_TCC_READ_REQ_sum"] / d["SQ_INSTS_VMEM_RD"], 3),
                   "clocks": int(clk), "cache_accesses_per_clk_per_cu": round(d["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / clk, 4)}
hit = [v["cache_accesses_per_clk_per_cu"] for k, v in l1_pmc.items() if k.endswith("_w1")]   # one block per CU: its 16 KiB stay in the 32 KiB L1
summary = {
    "session": tag, "source": "tests/microbench/peaks.hip on one MI355X (tests/microbench/run_peaks.sh): plain run, rocprofv3 --kernel-trace --stats, and --pmc passes of their own",
    "valu": {
        "unit": "SIMD cycles per wave64 VALU instruction at the 2.4 GHz maximum clock (1024 SIMDs x 2.4 GHz / measured wave-instructions per second), by waves per SIMD",
        "cycles_per_instr": valu,
        "traversal_mix_cycles_per_instr": mix["4"],
        "traversal_mix": "the slab test's own instructions -- 6 v_fma_f32, 3 v_max_f32, 3 v_min_f32, v_min3_f32, v_max3_f32, 2 v_cmp -- at four waves per SIMD, the path kernel's occupancy",
        "finding": "two classes: v_fma_f32 / v_add_f32 / v_mul_f32 / v_sub_f32 / v_and_b32 / v_add_u32 / v_mov_b32 issue every ~2.4 cycles once two or more waves share the SIMD (4.5-5.6 for one wave alone); "
                   "v_min / v_max / v_min3 / v_max3 / v_med3 / v_cmp / v_cndmask (e64) / v_lshlrev / v_cvt / v_fma_mix / v_pk_fma take ~4.1 cycles whatever the occupancy",
    },
    "l1": {
        "accesses_per_clk_per_cu": round(sum(hit) / len(hit), 3) if hit else None,
        "finding": "TCP_TOTAL_CACHE_ACCESSES counts one access per 128-B line an instruction touches (64 divergent lanes: 64; two lanes per line: 32) and one per 64 B of a fully coherent 1-KiB instruction (16); "
                   "the vector L1 retires 0.98-1.00 of them per clock and CU in every pattern, so 256 CUs x 2.4 GHz = 614 G accesses/s is the ceiling; four loads of one 64-B record cost four times one load",
        "by_pattern_plain_run": [{k: r[k] for k in ("pattern", "mode", "waves_per_simd", "ms", "G_load_instr_per_s", "clk_per_load_instr_per_cu", "B_per_clk_per_cu_requested")} for r in l1],
        "by_pattern_pmc": l1_pmc,
    },
}
json.dump(summary, open(os.path.join(out, f"peaks_{rnd}.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, f"{rnd}_peaks_kernel_stats.csv"))
for sub in ("pmc_l1", "pmc_valu"):
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        # the raw collection is ~100 KB: keep one line per (kernel, grid, counter)
        seen, keep = set(), []
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])
            if key not in seen:
                seen.add(key)
                keep.append({k: r[k] for k in ("Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value")})
        with open(os.path.join(out, f"{rnd}_peaks_{sub}.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
            w.writeheader()
            w.writerows(keep)
shutil.copy(os.path.join(src, "peaks.jsonl"), os.path.join(out, f"{rnd}_peaks.jsonl"))
print("traversal mix", mix, "L1 accesses/clk/CU", summary["l1"]["accesses_per_clk_per_cu"])
