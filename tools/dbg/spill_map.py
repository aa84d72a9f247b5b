"""Where a kernel spills: scratch loads / stores, matrix instructions and instruction counts per basic block of one function
of a hipcc -S listing.   python tools/dbg/spill_map.py file.s SUBSTRING_OF_THE_SYMBOL"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.split(':')[0] == pat or (l.startswith('_ZN') and pat in l.split(':')[0] and ':' in l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
cur = 'entry'; stats = {}; order = []
for i in range(start, end):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1)
    if cur not in stats:
        stats[cur] = {'n': 0, 'ld': 0, 'st': 0, 'mfma': 0, 'line': i, 'hdr': ''}; order.append(cur)
    if m and ';' in l:
        stats[cur]['hdr'] = l.split(';', 1)[1].strip()
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    stats[cur]['n'] += 1
    stats[cur]['ld'] += 'scratch_load' in t
    stats[cur]['st'] += 'scratch_store' in t
    stats[cur]['mfma'] += 'v_mfma' in t
tot = [0, 0, 0]
for c in order:
    s = stats[c]
    tot[0] += s['n']; tot[1] += s['ld']; tot[2] += s['st']
    if s['ld'] + s['st'] > 0 or s['mfma'] > 0 or s['n'] > 150:
        print('%-12s @%-6d instr %-5d ld %-4d st %-4d mfma %-4d %s' % (c, s['line'] - start, s['n'], s['ld'], s['st'], s['mfma'], s['hdr']))
print('total instr %d scratch ld %d st %d' % tuple(tot))
