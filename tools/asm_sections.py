#!/usr/bin/env python3
"""Static VALU / SALU / memory instruction count per source piece of one rt_path_kernel instantiation.
usage: tools/asm_sections.py <file.s from -save-temps -gline-tables-only> [mangled template args, default ILb1ELb0ELb1ELb1E]
Pieces are found by the source line of the lambda / function an instruction was inlined from."""
import re, collections, sys
src = open('/root/repo/raytracer_amd/csrc/rt_kernels.hip').read().split('\n')
def line_of(txt):
    for i, l in enumerate(src, 1):
        if txt in l: return i
    raise KeyError(txt)
MARKS = [(1, 'helpers (rng, sqrt_rn, div, unorm8)'), (line_of('template <bool CULL>'), 'phase1 (pk_fma)'), (line_of('__device__ __forceinline__ uint32_t combine_frame'), 'combine_frame'),
         (line_of('void rt_path_kernel(const rt_kargs a)'), 'prologue'),
         (line_of('auto own_sphere_drop'), 'own_sphere_drop'), (line_of('auto ray_prelude'), 'ray_prelude'), (line_of('auto test_sphere'), 'test_sphere (exact)'),
         (line_of('auto walk ='), 'walk loop'), (line_of('auto shade_miss'), 'shade_miss'), (line_of('auto shade_hit'), 'shade_hit'),
         (line_of('auto try_direction'), 'try_direction'), (line_of('auto start_ray'), 'start_ray'), (line_of('auto finish_direction'), 'finish_direction'),
         (line_of('auto trace_general'), 'trace_general'), (line_of('  for (;;) {'), '[E] + loop top'), (line_of('[F] frame end + [A] pixel hand-out'), '[F][A] service'),
         (line_of('// ================= [G] random draws, [C] trace + shade'), 'loop shapes (glue)')]
MARKS.sort()
def piece(line):
    cur = MARKS[0][1]
    for ln, tag in MARKS:
        if line >= ln: cur = tag
    return cur
kern = sys.argv[2] if len(sys.argv) > 2 else 'ILb1ELb0ELb1ELb1E'
inside = False; stack_line = 0
cnt = collections.Counter(); sal = collections.Counter(); mem = collections.Counter(); nop = collections.Counter()
for l in open(sys.argv[1]):
    if l.startswith('_ZN12_GLOBAL__N_114rt_path_kernel' + kern): inside = True; continue
    if inside and 's_endpgm' in l: break
    if not inside: continue
    m = re.match(r'\s*\.loc\s+1\s+(\d+)', l)
    if m: stack_line = int(m.group(1)); continue
    if re.match(r'\s*\.loc\s+\d+', l): continue
    m = re.match(r'\s*(v_\w+|s_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)', l)
    if not m: continue
    op = m.group(1); p = piece(stack_line)
    if op.startswith('v_'): cnt[p] += 1
    elif op == 's_nop': nop[p] += 1
    elif op.startswith('s_') and not re.match(r's_(load|buffer_load|waitcnt)', op): sal[p] += 1
    else: mem[p] += 1
print(kern, 'VALU', sum(cnt.values()), 'SALU', sum(sal.values()), 's_nop', sum(nop.values()), 'mem/wait', sum(mem.values()))
for _, tag in MARKS:
    print(f"   {tag:38s} VALU {cnt[tag]:5d}  SALU {sal[tag]:5d}  s_nop {nop[tag]:4d}  mem/wait {mem[tag]:4d}")
