#!/usr/bin/env python3
"""Instruction mix of a kernel in a hipcc -S listing: whole kernel and its hottest loop
(the basic blocks between the label that is the target of the last backward branch)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN3pnp\w+:', l) and pat in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('\t.section') or lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
def ops(ls):
    return [l.strip().split()[0] for l in ls if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
back = []
for i, l in enumerate(body):
    m = re.match(r'\s+s_cbranch\w*\s+(\.LBB\d+_\d+)|\s+s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            back.append((labels[tgt], i))
print('kernel', body[0][:60], 'instructions', len(ops(body)))
for lo, hi in sorted(back, key=lambda p: p[0] - p[1])[:3]:
    o = ops(body[lo:hi + 1])
    c = collections.Counter(o)
    print('loop lines %d..%d: %d instructions' % (lo, hi, len(o)))
    print('   ' + ', '.join('%s %d' % kv for kv in c.most_common(24)))
