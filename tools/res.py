#!/usr/bin/env python3
"""resource usage per kernel from a hipcc -S listing: tools_res.py file.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r"- \.agpr_count.*?\.wavefront_size", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if key not in name: continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk).group(1)
    print(name[:70], 'vgpr', g('vgpr_count'), 'vspill', g('vgpr_spill_count'), 'sgpr', g('sgpr_count'), 'sspill', g('sgpr_spill_count'), 'lds', g('group_segment_fixed_size'))
