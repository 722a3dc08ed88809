"""Ad-hoc: same library, different environment knobs, one subprocess each (interleaved twice)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1]
envs = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]]
code = open(os.path.join(ROOT, "tests", "gpu_variants.py")).read().split("code = r'''")[1].split("''' % (ROOT, wl)")[0] % (ROOT, wl)
for rnd in range(2):
    for e in envs:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **e), capture_output=True, text=True, timeout=280)
        print(e, r.stdout.strip() or r.stderr.strip()[-400:], flush=True)
