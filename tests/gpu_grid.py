import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = open(os.path.join(ROOT, "tests", "gpu_variants.py")).read().split("code = r'''")[1].split("''' % (ROOT, wl)")[0] % (ROOT, sys.argv[1] if len(sys.argv) > 1 else "atrium")
for m in (sys.argv[2:] or ["1", "2", "3", "4", "6", "8"]):
    env = dict(os.environ, RAYCA_GRID_MULT=m)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    print(f"grid x{m:3s}", r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
