#!/usr/bin/env python3
"""Where a kernel's VGPR pressure sits: highest VGPR index referenced per 100-line slice of its assembly (/tmp/f3d_fuse.s)."""
import collections, re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else 'k_fuseIdLb0E'
s = open('/tmp/f3d_fuse.s').read().split('\n')
start = [i for i, l in enumerate(s) if re.match(r'^_ZN\S*' + pat + r'\S*:', l)][0]
end = [i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm')][0]
body = s[start:end]
mx = []
for l in body:
    regs = [int(x) for x in re.findall(r'\bv(\d+)\b', l)] + [int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', l)]
    mx.append(max(regs) if regs else -1)
print('lines', len(body))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for k in range(0, len(body), step):
    print(k, max(mx[k:k + step]), [l.strip()[:60] for l in body[k:k + step] if l.strip().startswith(';') and 'f3d' in l][:2])
c = collections.Counter(l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';'))
print(c.most_common(30))
