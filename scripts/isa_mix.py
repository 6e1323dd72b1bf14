"""Instruction mix per kernel of a device assembly listing (hipcc -S --cuda-device-only).
usage: isa_mix.py file.s [name substring]"""
import collections, re, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:', s, re.S):
    name, body = m.group(1), m.group(2)
    if sub not in name:
        continue
    lines = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(l.split()[0] for l in lines)
    g = lambda p: sum(v for k, v in c.items() if re.match(p, k))
    short = re.sub(r'^_ZN2mh\d*(_GLOBAL__N_1)?\d*', '', name)[:40]
    print(f"{short:40s} instrs {len(lines):6d}  scratch {g('scratch_'):4d}  readlane {c['v_readlane_b32']:4d} writelane {c['v_writelane_b32']:4d}  "
          f"f64 {g(r'v_.*_f64'):5d}  ds {g('ds_'):4d}  global {g('global_'):4d}  waitcnt {c['s_waitcnt']:4d}  trans {g('v_(sin|cos|rcp|rsq|sqrt|exp|log)'):4d}  "
          f"branch {g('s_cbranch'):4d}  nop {c['s_nop']:4d}  dpp {sum(1 for l in lines if 'dpp' in l or 'row_' in l or 'quad_perm' in l):4d}")
