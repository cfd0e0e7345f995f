#!/usr/bin/env python3
"""Rough VGPR pressure profile of ONE kernel in a gfx950 .s file: straight-line live ranges (first definition .. last
use in program order, loops ignored) -> live count per basic block, and at the peak the source lines whose values are live.
    python tools/isa_pressure.py <file.s> <kernel-name-substring> [block label: breakdown at that block's peak]"""
import collections
import re
import sys

asm, kname = sys.argv[1], sys.argv[2]
lines = open(asm).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_][\w$.]*:', l) and kname in l.split(':')[0])
ins = []      # (block, srcline, defs, uses)
cur, loc = 'entry', 0


def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


for l in lines[start + 1:]:
    s = l.strip()
    if s.startswith('.Lfunc_end') or s.startswith('.amdhsa_kernel'):
        break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        loc = int(m.group(2))
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        cur = m.group(1)
        continue
    if not s or s[0] in '.;':
        continue
    s = s.split(';')[0]
    parts = s.split(None, 1)
    op = parts[0]
    ops = parts[1].split(',') if len(parts) > 1 else []
    if op.startswith('s_') and not any('v' in o for o in ops):
        continue
    store = op.startswith(('buffer_store', 'global_store', 'scratch_store', 'ds_write', 'ds_add', 'flat_store')) or \
        (op.startswith('buffer_load') and ' lds' in s) or op.startswith('v_cmp') and not op.startswith('v_cmpx') and False
    if store or op.startswith('v_cmp') or op.startswith('s_'):
        d, u = [], [r for o in ops for r in regs(o)]
    else:
        d = regs(ops[0]) if ops else []
        u = [r for o in ops[1:] for r in regs(o)]
        if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_writelane')) or 'dpp' in s and False:
            u += d
    ins.append((cur, loc, d, u))
first, last, defline = {}, {}, {}
# split into live ranges: a new definition of a register whose old value is not used afterwards starts a new range
ranges = []
open_rng = {}
for i, (_, ln, d, u) in enumerate(ins):
    for r in u:
        if r in open_rng:
            open_rng[r][1] = i
    for r in d:
        if r in open_rng:
            ranges.append(tuple(open_rng[r]))
        open_rng[r] = [i, i, ln, r]
for r in open_rng.values():
    ranges.append(tuple(r))
live = [0] * len(ins)
for a, b, _, _ in ranges:
    for i in range(a, b + 1):
        live[i] += 1
byblock = collections.OrderedDict()
for i, (blk, _, _, _) in enumerate(ins):
    byblock.setdefault(blk, []).append(live[i])
for blk, v in byblock.items():
    print("%-10s n=%4d  live max %3d  mean %5.1f" % (blk, len(v), max(v), sum(v) / len(v)))
want = sys.argv[3] if len(sys.argv) > 3 else None
peak = max((i for i in range(len(ins)) if want is None or ins[i][0] == want), key=lambda i: live[i])
print("peak %d live at instruction %d (block %s, source line %d); live values by defining source line:" % (live[peak], peak, ins[peak][0], ins[peak][1]))
c = collections.Counter()
for a, b, ln, r in ranges:
    if a <= peak <= b:
        c[ln] += 1
for ln, n in sorted(c.items()):
    print("   line %4d: %d" % (ln, n))
