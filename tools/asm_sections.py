#!/usr/bin/env python3
"""Static VALU / SALU / memory instruction count per kernel section (needs a -gline-tables-only .s from -save-temps)."""
import re, collections, sys
src = open('/root/repo/raytracer_amd/csrc/rt_kernels.hip').read().split('\n')
TAGS = ['[E] path end', '[F] frame end', '[A] hand out pixels', '[G] random draws', 'for (uint32_t round = 1',
        'if (state == ST_GOT_DIR)', '[C] trace + shade', 'phase 1: sign of D', 'phase 2: each lane', 'if (hit < 0) {',
        'const f32x4 sg = a.sph_shade']
marks = sorted((i, t) for i, l in enumerate(src, 1) for t in TAGS if t in l)
def section(line):
    cur = 'prologue'
    for ln, tag in marks:
        if line >= ln: cur = tag
    return cur
s = open(sys.argv[1]).read().split('\n')
for kern in ['ILb1ELb0ELb1E', 'ILb1ELb0ELb0E', 'ILb0ELb0ELb0E']:
    inside = False; line = 0; cnt = collections.Counter(); trans = collections.Counter(); sal = collections.Counter(); mem = collections.Counter()
    for l in s:
        if l.startswith('_ZN12_GLOBAL__N_114rt_path_kernel' + kern): inside = True; continue
        if inside and 's_endpgm' in l: break
        if not inside: continue
        m = re.match(r'\s*\.loc\s+\d+\s+(\d+)', l)
        if m: line = int(m.group(1)); continue
        m = re.match(r'\s*(s_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)', l)
        if m:
            if re.match(r's_(load|buffer_load|waitcnt|nop)', m.group(1)) or not m.group(1).startswith('s_'): mem[section(line)] += 1
            else: sal[section(line)] += 1
        m = re.match(r'\s*(v_\w+)', l)
        if m:
            cnt[section(line)] += 1
            if re.match(r'v_(sqrt|rcp|rsq|div_)', m.group(1)): trans[section(line)] += 1
    print(kern, 'VALU', sum(cnt.values()), 'SALU', sum(sal.values()), 'mem/wait', sum(mem.values()))
    for ln, tag in [(0, 'prologue')] + marks:
        print(f"   {tag:34s} VALU {cnt[tag]:5d} (trans/div {trans[tag]:3d})  SALU {sal[tag]:5d}  mem/wait {mem[tag]:4d}")
