"""Instruction histogram of one kernel from a hipcc -S dump: python tools/asm_stats.py file.s <substring> [out.s]"""
import collections
import re
import sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = None
for i, l in enumerate(s):
    if re.match(r'^_ZN\S*:', l) and key in l.split(':')[0]:
        start = i
        break
end = next(j for j in range(start, len(s)) if s[j].startswith('.Lfunc_end'))
body = s[start:end]
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write('\n'.join(body))
c = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l[0] in '.;' or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
print(s[start].split(':')[0], 'lines', len(body))
for k, v in c.most_common(40):
    print(f"{v:6d} {k}")
