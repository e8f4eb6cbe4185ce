"""Instruction census of the hottest loop (the backward branch that spans the most MFMAs) of every kernel in a hipcc -S listing.
usage: python tools/isa/loop_census.py file.s [substring of the mangled kernel name]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, m.group(1)) for i, l in enumerate(text) for m in [re.match(r'^(_Z\w+):\s', l)] if m]
for idx, (s0, name) in enumerate(starts):
    if want not in name:
        continue
    e0 = starts[idx + 1][0] if idx + 1 < len(starts) else len(text)
    lines = text[s0:e0]
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            s = labels[m.group(1)]
            n = sum('v_mfma' in x for x in lines[s:i])
            if best is None or n > best[2]:
                best = (s, i, n)
    if not best or best[2] == 0:
        continue
    s, e, n = best
    c = Counter()
    for l in lines[s:e + 1]:
        t = l.strip().split()
        if not t or t[0].startswith(';') or t[0].startswith('.'):
            continue
        c[t[0]] += 1
    tot = sum(c.values())
    print('%s\n  loop of %d lines: %d MFMA, %d other = %.2f per MFMA' % (name, e - s, n, tot - n, (tot - n) / n))
    print('  ' + ', '.join('%s:%d' % kv for kv in c.most_common(45)))
