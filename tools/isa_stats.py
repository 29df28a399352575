#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel in a hipcc -S listing whose name contains a pattern."""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    g = lambda k: re.search(k + r' (\d+)', body).group(1)
    print(name[:110], 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))
