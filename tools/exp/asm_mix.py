# instruction mix of kernels in rtx_kernels.s (make asm): asm_mix.py <mangled-name-prefix>...
import sys
from collections import Counter
s = open('/root/repo/rust-raytracer_amd/csrc/rtx_kernels.s').read().split('\n')
for name in sys.argv[1:]:
    on = False; ins = []
    for l in s:
        if l.startswith(name) and ':' in l: on = True; continue
        if on:
            t = l.strip()
            if t.startswith('s_endpgm'): break
            if l.startswith('\t') and t and not t.startswith(('.', ';')): ins.append(t)
    c = Counter(i.split()[0] for i in ins)
    g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print(name, len(ins), 'scratch', g('scratch_'), 'bperm', c.get('ds_bpermute_b32', 0), 'cndmask', g('v_cndmask'), 'ds_', g('ds_'), 'global', g('global_'), 'waitcnt', c.get('s_waitcnt', 0), 'f64', sum(v for k, v in c.items() if '_f64' in k))
