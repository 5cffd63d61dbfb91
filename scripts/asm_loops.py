"""Instruction mix per basic block of one kernel in a hipcc -S dump: python3 scripts/asm_loops.py /tmp/f3d_fuse.s <mangled-substring> [min_instrs]"""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end + 1]:
    t = l.strip()
    m = re.match(r'^(\.LBB\S+):', t)
    if m:
        blocks.append((name, cur)); cur, name = [], m.group(1); continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter()
    for op in ins:
        k = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') and not op.startswith('s_load') and not op.startswith('s_waitcnt') and not op.startswith('s_nop') else
             'smem' if op.startswith('s_load') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
        c[k] += 1; tot[k] += 1
        if op.startswith('v_readlane') or op.startswith('v_readfirstlane'): c['readlane'] += 1
        if op.startswith('v_pk_'): c['pk'] += 1
        if op.endswith('_f64') or '_f64_' in op: c['f64'] += 1
    if len(ins) >= minn:
        print(f'{name:14s} n={len(ins):4d} ' + ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
print('total', dict(tot))
