#!/usr/bin/env python3
"""Weighted VALU cycle estimate per basic block of a kernel's ISA (gfx950 rates measured by
tools/ubench/valu_rates.hip: simple VOP1/VOP2 and/or/add/sub/lshr/mov and f32 fma/mul/add issue in 2
cycles per wave64, everything else -- VOP3 integer, packed, dot, perm, cvt, SGPR-operand forms -- in 4)."""
import re, sys
from collections import Counter
FAST = {"v_and_b32","v_or_b32","v_add_u32","v_sub_u32","v_subrev_u32","v_lshrrev_b32","v_mov_b32","v_fma_f32","v_mul_f32",
        "v_add_f32","v_sub_f32","v_fmac_f32","v_xor_b32","v_not_b32","v_add_co_u32","v_addc_co_u32"}
def cost(line):
    op = line.split()[0]
    base = re.sub(r"_e32$|_e64$", "", op)
    if not op.startswith("v_"): return 0
    if base in FAST and "_e64" not in op and not re.search(r"\bs\d+|\bs\[", line.split(None,1)[1] if " " in line else ""):
        return 2
    return 4
def main(path, kernel):
    s = open(path).read()
    a = s.index(kernel); a = s.index(":", a); b = s.index(".Lfunc_end", a)
    blocks = []; cur = ["entry", []]; blocks.append(cur)
    for l in s[a:b].split("\n"):
        ls = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", ls) or re.match(r"^; %bb\.(\d+):", ls)
        if m: cur = [ls[:48], []]; blocks.append(cur); continue
        if not ls or ls[0] in ";.": continue
        cur[1].append(ls)
    tot = 0
    for name, ins in blocks:
        c = sum(cost(i) for i in ins); n = sum(1 for i in ins if i.startswith("v_"))
        sal = sum(1 for i in ins if i.startswith("s_"))
        if n + sal > 4:
            print("%-50s valu %3d  cycles %4d  salu %3d  %s" % (name, n, c, sal, "LOOP" if "Loop" in name else ""))
main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "_ZN3p2p3w6418remap_views_kernelE")
