"""Condense tools/r3_ab.sh output (stdin) into one line per (config, mode): Grays/s per library and repetition."""
import sys, re, collections
d = collections.defaultdict(list)
for l in sys.stdin:
    m = re.match(r'(\S+)\s+(C\d) (\S+)\s+n=\d+\s+[\d.]+ ms\s+([\d.]+) Grays', l)
    if m:
        d[(m.group(2), m.group(3))].append((m.group(1), float(m.group(4))))
    elif 'passed' in l or 'failed' in l or 'equal False' in l:
        print(l.rstrip())
for k in sorted(d):
    acc = collections.defaultdict(list)
    for t, v in d[k]:
        acc[t].append(v)
    print("%-3s %-9s " % k + '   '.join('%s %s' % (t, '/'.join('%.1f' % x for x in v)) for t, v in acc.items()))
