#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in a gfx950 assembly file produced with
`hipcc --save-temps` (reads the amdhsa.kernels metadata). Spills mean scratch traffic through L2 to HBM."""
import re
import sys

txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    f = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    name = f.get("name", "?")
    m = re.search(r"fwht_pass_kernelI(\w)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
    if m:
        t, K, logv, rb, lu, llds, inm, outm, single = m.groups()
        name = "fwht<%s K=%s tile=(%s,%s,%s,%s) in=%s out=%s single=%s>" % ({"d": "f64", "f": "f32"}[t], K, logv, rb, lu, llds, inm, outm, single)
    print("%-62s vgpr %3s agpr %3s spill %3s scratch %4s B lds %6s" % (name[:62], f.get("vgpr_count"), f.get("agpr_count"),
          f.get("vgpr_spill_count"), f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
