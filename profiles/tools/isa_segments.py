"""Static per-barrier-segment instruction mix of one kernel in an asm file: segs.py file.s kernel_substring"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().endswith(('E:', 'E')) or (l.startswith('_ZN') and key in l and ':' in l))
seg = {'mfma': 0, 'trans': 0, 'valu': 0, 'ds': 0, 'vmem': 0, 'salu': 0, 'wait': 0, 'lane': 0, 'scratch': 0}
def flush(tag, idx):
    est = seg['mfma'] * 16 and 0
    print("%5d %-14s mfma %3d trans %3d valu %4d lane %3d ds %3d vmem %3d scr %2d salu %4d wait %3d | valu-issue~%5d mfma~%5d" % (
        idx, tag, seg['mfma'], seg['trans'], seg['valu'], seg['lane'], seg['ds'], seg['vmem'], seg['scratch'], seg['salu'], seg['wait'],
        4 * seg['valu'] + 8 * seg['trans'] + 4 * seg['lane'], seg['mfmac']))
    for k in seg: seg[k] = 0
seg['mfmac'] = 0
n = 0
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith('.Lfunc_end') or l.startswith('s_endpgm'):
        flush('end', i); break
    if not l or l.startswith(';') or l.startswith('.'):
        if l.startswith('.LBB'):
            flush('label ' + l.split(':')[0], i)
        continue
    op = l.split()[0]
    if op == 's_barrier':
        flush('barrier', i); continue
    if op.startswith('v_mfma'):
        seg['mfma'] += 1; seg['mfmac'] += 32 if 'x4_f32' in op or '16x16x4f32' in op else 16
    elif re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op): seg['trans'] += 1
    elif op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): seg['lane'] += 1
    elif op.startswith('v_'): seg['valu'] += 1
    elif op.startswith('ds_'): seg['ds'] += 1
    elif op.startswith('scratch_'): seg['scratch'] += 1
    elif op.startswith(('buffer_', 'global_', 'flat_')): seg['vmem'] += 1
    elif op == 's_waitcnt': seg['wait'] += 1
    elif op.startswith('s_'): seg['salu'] += 1
