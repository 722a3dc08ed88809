"""Summarise rocprofv3 --pmc CSVs: per kernel, per counter, mean over dispatches."""
import csv, glob, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_generation" in k: k = "k_generation"
        elif "k_flat_refill" in k: k = "k_flat_refill"
        elif "k_resolve" in k: k = "k_resolve"
        else: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in out.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:36s} {v:18.1f}")
json.dump(out, open(root + "/summary.json", "w"), indent=1)
