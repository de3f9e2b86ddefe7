#!/usr/bin/env python3
"""VGPR / occupancy / scratch of every decode kernel instantiation in a hipcc -save-temps .s file.
usage: tools/kernel_regs.py <file.s> [name-filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2:] or ["k_qkv", "k_ffn13", "k_gemv_res", "k_cls"]
blocks = re.split(r'^(_Z[\w]+):\s*; @.*$', txt, flags=re.M)
rows = []
for i in range(1, len(blocks), 2):
    n, body = blocks[i], blocks[i + 1]
    m = re.search(r'; NumVgprs: (\d+)', body)
    if not m or not any(f in n for f in flt):
        continue
    o = re.search(r'; Occupancy: (\d+)', body)
    s = re.search(r'; ScratchSize: (\d+)', body)
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    rows.append((d.split("(")[0].replace("void ", ""), int(m.group(1)), int(o.group(1)), int(s.group(1))))
for r in sorted(rows):
    print("%-40s vgprs %3d  waves/SIMD %d  scratch %d" % r)
