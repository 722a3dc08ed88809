"""Turn the outlier counts of a GPU run (gpurun_out/parity_outliers.json, written by tests/parity_report.py under
RAYCA_PARITY_MEASURE=1) into tests/parity_bounds.json: bound = measured count + max(2, 25 %), zero stays zero."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rep = json.load(open(os.path.join(ROOT, "gpurun_out", "parity_outliers.json")))
out = {}
for name, r in sorted(rep.items()):
    n = int(r["pixels_beyond_tolerance"])
    if n == 0:
        continue
    out[name] = {"measured": n, "pixels": r["pixels"], "worst": r["worst"], "bound": n + max(2, (n + 3) // 4)}
json.dump(out, open(os.path.join(ROOT, "tests", "parity_bounds.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
