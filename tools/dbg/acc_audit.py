"""wrmf_chol_mf.hip keeps its accumulators in a0..a159 BY NAME (inline asm) and gathers with asm loads that hipcc does not
count; two things must then hold in the listing hipcc produces (build.py checks them at every build):
  1. no compiler-generated instruction touches the accumulator file (hipcc spills there when a kernel uses it -- the forced
     function attribute amdgpu-agpr-alloc=0 is what stops it);
  2. no compiler-generated instruction reads or writes a register that an asm load has in flight, i.e. between the
     ;;#ASMSTART global_load_dword vN ... and the next asm s_waitcnt vmcnt(0) (a copy or a spill of such a register
     moves data that has not landed).
Scratch instructions are listed too (allowed, but every one of them is a cost).
    python tools/dbg/acc_audit.py file.s [-q]   -> exit status 1 if rule 1 or 2 is broken"""
import re, sys


def regs_of(text):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', text):
        out.add(int(m.group(1)))
    return out


def audit(path, quiet=False, acc_floor=0):
    """acc_floor: only compiler-generated accesses to a[acc_floor] and above break rule 1 (-DMF_SAFE builds: the tiles start there)"""
    lines = open(path).read().split('\n')
    # split into functions, functions into basic blocks (a label line starts one; so does the line after a branch)
    funcs, cur_fn = {}, None
    for l in lines:
        if re.match(r'^[A-Za-z_][\w.$]*:', l) and not l.startswith('.L'):
            cur_fn = l.split(':')[0]; funcs[cur_fn] = []
        elif l.startswith('.Lfunc_end'):
            cur_fn = None
        elif cur_fn is not None:
            funcs[cur_fn].append(l)
    acc, flight_hits, scratch = [], [], 0
    for fn, body in funcs.items():
        blocks, order, name, inasm = {}, [], 'entry', False
        blocks[name] = []; order.append(name)
        k = 0
        for l in body:
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                name = m.group(1); blocks[name] = []; order.append(name); continue
            if '#ASMSTART' in l: inasm = True; continue
            if '#ASMEND' in l: inasm = False; continue
            t = l.split(';')[0].strip()
            if not t or t.startswith('.'):
                continue
            blocks[name].append((inasm, t))
            if not inasm and (t.startswith('s_branch') or t.startswith('s_cbranch')):
                k += 1; name = '%s.after%d' % (order[-1], k); blocks[name] = []; order.append(name)
        succ = {}
        for i, b in enumerate(order):
            ins = blocks[b]
            out = []
            last = ins[-1][1] if ins else ''
            if last.startswith('s_cbranch') or last.startswith('s_branch'):
                out.append(last.split()[-1])
            if not last.startswith('s_branch') and not last.startswith('s_endpgm') and i + 1 < len(order):
                out.append(order[i + 1])
            succ[b] = [o for o in out if o in blocks]

        def transfer(b, state, report):
            st = set(state)
            for inasm_, t in blocks[b]:
                if inasm_:
                    if t.startswith('global_load_dword'):
                        st |= regs_of(t.split(',')[0])
                    elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                        st = set()
                    continue
                if report:
                    if 'accvgpr' in t or re.search(r'\ba\[?\d', t):
                        idx = [int(x) for x in re.findall(r'\ba\[?(\d+)', t)] + [int(x) for x in re.findall(r'\ba\[\d+:(\d+)\]', t)]
                        if not idx or max(idx) >= acc_floor:
                            acc.append((fn[-44:], b, t))
                    if st and (regs_of(t) & st):
                        flight_hits.append((fn[-44:], b, t))
            return st
        instate = {b: set() for b in order}
        changed = True
        while changed:
            changed = False
            for b in order:
                o = transfer(b, instate[b], False)
                for s2 in succ[b]:
                    if not o <= instate[s2]:
                        instate[s2] |= o; changed = True
        for b in order:
            transfer(b, instate[b], True)
            scratch += sum(1 for ia, t in blocks[b] if not ia and 'scratch_' in t)
    if not quiet:
        for fn, cur, t in acc[:20]: print('ACC   ', fn, cur, t)
        for fn, cur, t in flight_hits[:20]: print('FLIGHT', fn, cur, t)
    print('compiler-generated accumulator-file instructions: %d, touching a register in flight: %d, scratch instructions: %d'
          % (len(acc), len(flight_hits), scratch))
    return len(acc), len(flight_hits), scratch


if __name__ == '__main__':
    a, f, _ = audit(sys.argv[1], '-q' in sys.argv)
    sys.exit(1 if (a or f) else 0)
