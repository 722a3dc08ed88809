"""Summarise rocprofv3 --pmc CSVs: per kernel class, per counter, mean over dispatches; the dispatches' own duration
(End_Timestamp - Start_Timestamp of the collection records) goes in as launch_ms_under_pmc.
usage: python tests/pmc_summary.py gpurun_out/pmc_<tag>"""
import csv, glob, sys, collections, json
CLASSES = ("k_generation", "k_flat_refill", "k_queue_refill", "k_shadow_refill", "k_wf_trace", "k_wf_shade", "k_wf_shadow", "k_resolve", "k_general")
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
full = {}
for f in sorted(glob.glob(root + "/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = next((c for c in CLASSES if c + "<" in r["Kernel_Name"] or c + "(" in r["Kernel_Name"]), None)
        if k is None:
            continue
        full.setdefault(k, r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][(f, r["Dispatch_Id"])] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k in out:
    d = list(dur[k].values())
    out[k]["launch_ms_under_pmc"] = sum(d) / len(d)
    out[k]["dispatches_averaged"] = len(d)
    out[k]["kernel_name"] = full[k]
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} {v:18.4f}" if not isinstance(v, str) else f"   {c:36s} {v[:120]}")
json.dump(out, open(root + "/summary.json", "w"), indent=1)
