#!/usr/bin/env python3
"""Basic blocks of ONE kernel in a gfx950 .s file (compiled with -gline-tables-only) with instruction-class counts and
the source lines they come from -- the static side of a per-phase VALU budget.

    python tools/isa_blocks.py <file.s> <kernel-name-substring> [source.hip]

Prints one row per basic block: label, VALU / packed-FMA / DPP-mov / SALU / LDS / VMEM counts, branch target, and the
three source lines (of the main file) that contribute most VALU instructions.  Weights for a dynamic estimate are the
reader's: a block inside the round loop runs once per wave and round, a far step once per wave, round and step."""
import collections
import re
import sys


def main():
    asm, kname = sys.argv[1], sys.argv[2]
    src = open(sys.argv[3]).read().split('\n') if len(sys.argv) > 3 else None
    lines = open(asm).read().split('\n')
    # file table: .file N "dir" "name"
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[A-Za-z_][\w$.]*:', l) and kname in l.split(':')[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []
    cur = {'label': 'entry', 'c': collections.Counter(), 'src': collections.Counter(), 'br': []}
    curloc = None
    mainfile = None
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end') or s.startswith('.section') or s.startswith('.amdhsa_kernel'):
            break
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            curloc = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r'^\.LBB\d+_\d+:', s):
            blocks.append(cur)
            cur = {'label': s[:-1].split(':')[0], 'c': collections.Counter(), 'src': collections.Counter(), 'br': []}
            continue
        if not s or s.startswith('.') or s.startswith(';'):
            continue
        op = s.split()[0]
        c = cur['c']
        if op.startswith('v_'):
            c['valu'] += 1
            if op.startswith('v_pk_fma') or op.startswith('v_pk_mul') or op.startswith('v_pk_add'):
                c['pk'] += 1
            if '_dpp' in s or 'quad_perm' in s or 'row_shr' in s:
                c['dpp'] += 1
            if curloc:
                cur['src'][curloc] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
            if op.startswith('s_cbranch') or op == 's_branch':
                cur['br'].append(s.split()[-1])
            if op.startswith('s_waitcnt'):
                c['wait'] += 1
            if op == 's_barrier':
                c['barrier'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'):
            c['vmem'] += 1
            if op.startswith('scratch_'):
                c['scratch'] += 1
    blocks.append(cur)
    tot = collections.Counter()
    print("%-12s %5s %4s %4s %5s %4s %4s  %-18s %s" % ("block", "valu", "pk", "dpp", "salu", "lds", "vmem", "branches", "top source lines (valu)"))
    for b in blocks:
        c = b['c']
        tot.update(c)
        top = []
        for (f, ln), n in b['src'].most_common(3):
            fn = files.get(f, '?')
            tag = "%d" % ln if fn.endswith('.hip') else "%s:%d" % (fn.split('/')[-1][:12], ln)
            top.append("%s(%d)" % (tag, n))
        print("%-12s %5d %4d %4d %5d %4d %4d  %-18s %s" % (b['label'][:12], c['valu'], c['pk'], c['dpp'], c['salu'], c['lds'], c['vmem'],
                                                      ','.join(x.replace('.LBB', 'B') for x in b['br'])[:18], ' '.join(top)))
    print("total", dict(tot))


if __name__ == '__main__':
    main()
