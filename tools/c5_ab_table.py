"""tools/c5_ab.sh output -> one table: per size and kernel, the average us of every run of every spec"""
import collections
import re
import sys
d = collections.defaultdict(list); order = []; cur = None
for l in open(sys.argv[1]):
    m = re.match(r'=== (\S+) (\d) (\w+) (.*)', l)
    if m:
        cur = (m.group(1), m.group(3)); d[(cur, 'rate')].append(m.group(4).strip().split('|')[-1].split()[0])
        if m.group(1) not in order: order.append(m.group(1))
        continue
    m = re.match(r'(?:akmi::)?(\S+(?:, \S+)*)\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
    if m and cur: d[(cur, m.group(1))].append(float(m.group(4)))
for sz in ('deck', 'prod'):
    print('##', sz)
    for k in sorted(set(k[1] for k in d if k[0][1] == sz)):
        print('%-42s' % k, ' | '.join('%s %s' % (v, d[((v, sz), k)]) for v in order))
