"""Instruction-class pattern of a line range of an asm file: isa_pattern.py file.s first_line last_line
M = matrix, v = vector ALU, t = transcendental, d = LDS, g = global / buffer, s = scalar, w = s_waitcnt, B = barrier, x = scratch"""
import re, sys
path, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
out = []
for l in open(path).read().split('\n')[lo:hi]:
    l = l.strip()
    if not l or l.startswith((';', '.')):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif re.match(r'v_(exp|log|rcp|rsq|sqrt)_', op): out.append('t')
    elif op.startswith('v_'): out.append('v')
    elif op.startswith('ds_'): out.append('d')
    elif op.startswith('scratch_'): out.append('x')
    elif op.startswith(('buffer_', 'global_')): out.append('g')
    elif op == 's_waitcnt': out.append('w')
    elif op == 's_barrier': out.append('B')
    elif op == 's_nop': out.append('n')
    elif op.startswith('s_'): out.append('s')
s = ''.join(out)
for i in range(0, len(s), 150):
    print(s[i:i + 150])
