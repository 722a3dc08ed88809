"""Register / spill / scratch / LDS figures of every kernel, read from the gfx950 code object's metadata.
usage: python tests/kernel_resources.py [extra hipcc flags ...]   (compiles rayca_amd/csrc/kernels.hip device-only into /tmp;
RAYCA_KRES_UNIT=refill|bvh_build picks another translation unit)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
out = "/tmp/rayca_kernels_dev.o"
src = os.path.join(g.CSRC, os.environ.get("RAYCA_KRES_UNIT", "kernels") + ".hip")
out = f"/tmp/rayca_{os.environ.get('RAYCA_KRES_UNIT', 'kernels')}_dev.o"
if not os.environ.get("RAYCA_KRES_REUSE"):
    subprocess.run([g.HIPCC, "--offload-arch=gfx950", "--cuda-device-only", *g.COMMON, *sys.argv[1:], "-c", src, "-o", out], check=True)
co = out + ".co"   # the device-only object is still an offload bundle: take the gfx950 code object out of it
subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + out, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                "--output=" + co], check=True)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
rows = []
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    f = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").replace("void rayca::", "").replace("rayca::", "")
    dem = re.sub(r"\(.*", "", dem)
    rows.append((dem, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"), f("private_segment_fixed_size")))
print(f"{'kernel':64s} vgpr vspill sgpr sspill scratch_B")
for r in sorted(rows):
    print(f"{r[0]:64s} {r[1]:4d} {r[2]:6d} {r[3]:4d} {r[4]:6d} {r[5]:9d}")
