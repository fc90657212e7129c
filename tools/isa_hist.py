"""Histogram of the instructions inside every loop of a kernel's ISA listing (hipcc -S --cuda-device-only).
usage: python tools/isa_hist.py kernel.s"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))
def cat(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('ds_'): return 'DS:' + op
    if op.startswith('v_'): return 'V:' + op
    if op.startswith('s_'): return 'S'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'VMEM:' + op
    return op
for a, b in loops:
    c = collections.Counter()
    for l in lines[a:b + 1]:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if m: c[cat(m.group(1))] += 1
    tot = sum(v for k, v in c.items() if k.startswith('V:'))
    print('loop lines %d-%d: MFMA %d VALU %d DS %d SALU %d VMEM %d' % (a, b, c['MFMA'], tot, sum(v for k, v in c.items() if k.startswith('DS')), c['S'],
                                                                    sum(v for k, v in c.items() if k.startswith('VMEM'))))
    print('   ' + ', '.join('%s %d' % (k[2:] if k[1] == ':' else k, v) for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k[:2] in ('V:', 'DS', 'VM')))
