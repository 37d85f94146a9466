#!/usr/bin/env python3
"""Developer tool: executed-path instruction census of the tile kernel (tools/census.hip).
usage: tools/census.py [-v]   -> per kernel: VALU count, weighted issue slots (tools/instr_rate.hip weights),
s_nop wait states, VGPRs; per coefficient = / 16."""
import os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32',
        'v_lshrrev_b32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'}
def main():
    out = '/tmp/census.s'
    extra = (['-DRONK_CENSUS_BREAKDOWN'] if '--breakdown' in sys.argv else []) + (['-DRONK_CENSUS_ALL_LAZY'] if '--all-lazy' in sys.argv else [])
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only'] + extra +
                          ['-I' + os.path.join(ROOT, 'include'), '-o', out, os.path.join(ROOT, 'tools/census.hip')],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    tot = {}
    for m in re.finditer(r'^(_Z\w*census_kernel\w+):.*?s_endpgm(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
        name = m.group(1)
        body = m.group(0)
        tag = re.search(r'ILi(\d+)ELb(\d)ELi(\d)ELi(\d+)', name).groups()
        ins = [l.split() for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        c = Counter(i[0] for i in ins)
        valu = sum(n for k, n in c.items() if k.startswith('v_'))
        slots = sum((1.0 if re.sub(r'_e(32|64)$', '', k) in FAST else 1.6) * n for k, n in c.items() if k.startswith('v_'))
        nops = sum(int(i[1]) + 1 for i in ins if i[0] == 's_nop')
        vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
        sp = re.search(r'; ScratchSize: (\d+)', s[m.end():m.end() + 3000])
        print('R=2^%s mode %s abl %s: VALU %5d (%.1f/coef)  slots %7.1f (%.1f/coef)  s_nop states %4d  mov %d  vgpr %s scratch %s' % (
            tag[0], tag[2], tag[3], valu, valu / 16, slots, slots / 16, nops, c.get('v_mov_b32_e32', 0), vg, sp.group(1) if sp else '?'))
        if tag[3] != '0':
            continue                       # breakdown instantiations: listed, not summed
        # a transform = one column pass + one row pass: 'two-level' sums modes 0 + 1, 'matrix' sums modes 3 + 1
        for variant, modes in (('two-level', '01'), ('matrix', '31')):
            if tag[2] in modes:
                t_ = tot.setdefault((tag[0], variant), [0, 0.0])
                t_[0] += valu; t_[1] += slots
        if '-v' in sys.argv:
            print('   ', dict(c.most_common(25)))
    # The kernels the library runs at 2^22 since round 6: ntt_tile_wl.h, 2^11 rows x 4 columns, FULL image (everything is a template
    # parameter there, so the product's own translation unit is the executed path: no census instantiation needed)
    out_wl = '/tmp/census_wl.s'
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                           '-I' + os.path.join(ROOT, 'include'), '-o', out_wl, os.path.join(ROOT, 'ronkathon_amd/csrc/tile_kernels_wl.hip')],
                          stderr=subprocess.DEVNULL)
    sw = open(out_wl).read()
    wl = {}
    for label, sym in (('column/matrix', r'_ZN4ronk22ntt_tile_wl_col_kernelILi11ELb0ELi3ELb1ENS_7GlFieldEEEvNS_8TileArgsE'),
                       ('column/two-level', r'_ZN4ronk22ntt_tile_wl_col_kernelILi11ELb0ELi1ELb1ENS_7GlFieldEEEvNS_8TileArgsE'),
                       ('row', r'_ZN4ronk22ntt_tile_wl_row_kernelILi11ELb0ELb1ENS_7GlFieldEEEvNS_8TileArgsE')):
        m = re.search(r'^' + sym + r':.*?s_endpgm', sw, re.S | re.M)
        ins = [l.split() for l in m.group(0).splitlines() if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        c = Counter(i[0] for i in ins)
        valu = sum(n for k, n in c.items() if k.startswith('v_'))
        slots = sum((1.0 if re.sub(r'_e(32|64)$', '', k) in FAST else 1.6) * n for k, n in c.items() if k.startswith('v_'))
        wl[label] = (valu, slots)
        print('wave-local 2^11 x 4 %-17s: VALU %5d (%.1f/coef)  slots %7.1f (%.1f/coef)' % (label, valu, valu / 16, slots, slots / 16))
    for variant in ('matrix', 'two-level'):
        v = wl['column/' + variant][0] + wl['row'][0]; sl = wl['column/' + variant][1] + wl['row'][1]
        print('transform with 2^11 wave-local passes, inter-pass twiddle %-9s: VALU/coef %.1f  slots/coef %.1f' % (variant, v / 16, sl / 16))
    for (k, variant), (v, sl) in sorted(tot.items()):
        print('transform with 2^%s passes, inter-pass twiddle %-9s: VALU/coef %.1f  slots/coef %.1f' % (k, variant, v / 16, sl / 16))
    if '--write' in sys.argv:   # profiles/latest_census.json: bench.py's roofline.valu (secondary, VALU-issue ceiling)
        import json
        sys.path.insert(0, ROOT)
        import bench
        v = wl['column/matrix'][0] + wl['row'][0]; sl = wl['column/matrix'][1] + wl['row'][1]
        v2 = wl['column/two-level'][0] + wl['row'][0]; sl2 = wl['column/two-level'][1] + wl['row'][1]
        with open(os.path.join(ROOT, 'profiles', 'latest_census.json'), 'w') as f:
            json.dump({'kernel_source_hash': bench.kernel_source_hash(),
                       'workload': 'ntt22 (two 2^11-row passes of 4-column tiles, ntt_tile_wl.h, full inter-pass twiddle matrix: the library default at 2^22)',
                       'two_level_variant': {'valu_per_coeff': v2 / 16, 'slots_per_coeff': sl2 / 16},
                       'ntt_tile_h_c8': {'valu_per_coeff': tot[('11', 'matrix')][0] / 16, 'slots_per_coeff': tot[('11', 'matrix')][1] / 16},
                       'valu_per_coeff': v / 16, 'slots_per_coeff': sl / 16,
                       'how': 'tools/census.py: static count of the executed path of tools/census.hip (plan flags fixed at '
                              'compile time), weights 1.0 (simple 32-bit VALU) / 1.6 (everything else) from tools/instr_rate.hip'},
                      f, indent=1)
if __name__ == '__main__':
    main()
