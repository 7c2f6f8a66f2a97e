"""List the memory operations, waits and branches of one kernel in a hipcc -S listing: asm_waits.py file.s name-fragment [index]"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\S*' + sys.argv[2] + r'\S*):', s, re.M)]
n = names[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
start = s.index(n + ':')
body = s[start:s.index('.Lfunc_end', start)].split('\n')
print(n, len(body), 'lines')
for k, l in enumerate(body):
    if any(w in l for w in ('s_waitcnt', 'global_load', 'global_store', 's_cbranch', 's_branch', 's_barrier', 'ds_read', 'ds_write', 'scratch_')) or l.startswith('.LBB'):
        print(k, l.strip()[:90])
