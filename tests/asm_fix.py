"""Post-pass over the gfx950 assembly of a kernel translation unit (experiment, tests/build_asm_variant.sh):
  cndmask   v_cndmask_b32_e32 vD, vA, vB, vcc  ->  v_cndmask_b32_e64 vD, vA, vB, vcc
            (tests/microbench/peaks.hip: the VOP2 form with its implicit vcc issues every ~23 cycles back to back on this
             chip, the VOP3 form with the mask named as an SGPR pair every ~4)
usage: python tests/asm_fix.py in.s out.s [cndmask]"""
import re, sys
src, dst, what = sys.argv[1], sys.argv[2], set(sys.argv[3:]) or {"cndmask"}
n = 0
out = []
for line in open(src):
    if "cndmask" in what:
        m = re.match(r"^(\s*)v_cndmask_b32_e32 (v\d+), ([^,]+), (v\d+), vcc\s*$", line)
        if m:
            line = f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {m.group(3)}, {m.group(4)}, vcc\n"
            n += 1
    out.append(line)
open(dst, "w").writelines(out)
print(f"asm_fix: {n} instructions rewritten", file=sys.stderr)
