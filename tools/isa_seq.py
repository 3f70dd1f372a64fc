# Memory-operation skeleton of one kernel in the device assembly: loads (L), stores (S), atomics (A), waits (Wn),
# branches (|). Usage: hipcc ... --cuda-device-only -S -o /tmp/isa/ptgpu.s ...; python tools/isa_seq.py /tmp/isa/ptgpu.s 'k_shade_guidedILi3E'
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if l.startswith('_ZN') and key in l and l.rstrip().endswith(l.split(':')[0][1:]) or (l.startswith('_ZN') and key in l and ':' in l))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
seq = []
for ln in s[start:end]:
    t = ln.strip()
    op = t.split()[0] if t else ''
    if op.startswith('global_load') or op.startswith('buffer_load'): seq.append('L' + op.split('_')[-1].replace('dwordx', 'x').replace('dword', 'x1'))
    elif op.startswith('scratch_load'): seq.append('scrL')
    elif op.startswith('scratch_store'): seq.append('scrS')
    elif op.startswith('global_store'): seq.append('S' + op.split('_')[-1].replace('dwordx', 'x').replace('dword', 'x1'))
    elif op.startswith('global_atomic'): seq.append('A')
    elif op.startswith('ds_'): seq.append('d')
    elif op == 's_waitcnt' and 'vmcnt' in t: seq.append('W' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    elif op.startswith('s_cbranch') or op == 's_branch': seq.append('|')
    elif op == 's_barrier': seq.append('BAR')
    elif op.startswith('s_swappc') or op.startswith('s_setpc'): seq.append('CALL')
out = []; prev = None; cnt = 0
for x in seq + [None]:
    if x == prev: cnt += 1
    else:
        if prev: out.append(prev + (f'*{cnt}' if cnt > 1 else ''))
        prev = x; cnt = 1
print(end - start, 'lines')
print(' '.join(out))
