#!/usr/bin/env python3
"""Largest scratch offset and number of scratch instructions per symbol of the gfx950 code objects in the library
(static, from the ISA): which function's frame makes a kernel's bytes-per-lane.  usage: tools/scratch_by_symbol.py lib.so [regex]"""
import collections
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_histogram as ih

lib = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "BlsCurve"
with tempfile.TemporaryDirectory() as tmp:
    rows = []
    for co in ih.extract_code_objects(os.path.abspath(lib), tmp):
        syms = ih.disassemble(co)
        dm = ih.demangle(list(syms))
        for s, insts in syms.items():
            if not re.search(pat, dm.get(s, s)):
                continue
            mx, cnt = 0, 0
            for _, op, args in insts:
                if op.startswith("scratch_"):
                    cnt += 1
                    m = re.search(r"offset:(\d+)", args)
                    if m:
                        mx = max(mx, int(m.group(1)))
            if cnt:
                rows.append((mx, cnt, len(insts), re.sub(r"\(.*", "", dm.get(s, s))[:110]))
    for r in sorted(rows, reverse=True)[:40]:
        print("max_offset=%5d scratch_ops=%5d insts=%7d  %s" % r)
