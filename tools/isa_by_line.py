#!/usr/bin/env python3
"""VALU / LDS / VMEM instruction counts per source line of a kernel compiled with -gline-tables-only (static counts):
    python tools/isa_by_line.py <file.s> <source.hip> [min_count]"""
import collections
import re
import sys

asm, srcf = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cur = None
cnt = collections.defaultdict(collections.Counter)
for l in open(asm):
    l = l.strip()
    m = re.match(r'\.loc\s+\d+\s+(\d+)', l)
    if m:
        cur = int(m.group(1))
        continue
    if re.match(r'^v_', l):
        cnt[cur]['valu'] += 1
    elif re.match(r'^ds_', l):
        cnt[cur]['lds'] += 1
    elif re.match(r'^(buffer_|global_|scratch_)', l):
        cnt[cur]['vmem'] += 1
src = open(srcf).read().split('\n')
for ln in sorted(k for k in cnt if k):
    c = cnt[ln]
    if c['valu'] + c['lds'] + c['vmem'] >= minc:
        print("%4d valu %4d lds %3d vmem %3d  %s" % (ln, c['valu'], c['lds'], c['vmem'], src[ln - 1].strip()[:100]))
print("total valu", sum(c['valu'] for c in cnt.values()))
