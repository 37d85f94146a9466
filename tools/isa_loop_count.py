#!/usr/bin/env python3
"""Developer tool: weighted instruction count of the hottest loop of every kernel in a gfx950 .s file.
Weights from tools/instr_rate.hip on MI355X: simple 32-bit VALU 1.0, other VALU 1.6, s_nop 0.3.
usage: isa_loop_count.py file.s [ops_per_iteration]"""
import re
import sys
from collections import Counter

FAST = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_not_b32', 'v_mov_b32',
        'v_lshrrev_b32', 'v_accvgpr_write_b32', 'v_accvgpr_read_b32'}


def weight(op):
    base = re.sub(r'_e(32|64)$', '', op)
    if base.startswith('v_'):
        return 1.0 if base in FAST else 1.6
    if base == 's_nop':
        return 0.3
    return 0.0


def main():
    s = open(sys.argv[1]).read()
    per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    for m in re.finditer(r'^(\w+):\s*; @\1\n(.*?)s_endpgm', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = body.splitlines()
        labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
        best = None
        for i, l in enumerate(lines):
            mm = re.match(r'\s+s_cbranch_\w+ (\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = lines[labels[mm.group(1)]:i + 1]
                if best is None or len(seg) > len(best):
                    best = seg
        if not best:
            best = lines
        ins = [l.split()[0] for l in best if l.startswith('\t') and not l.strip().startswith((';', '.'))]
        c = Counter(ins)
        v = sum(n for k, n in c.items() if k.startswith('v_'))
        w = sum(weight(k) * n for k, n in c.items())
        print('%s: VALU/op %.2f  slots/op %.2f' % (name, v / per, w / per))
        print('   ', dict(c.most_common(12)))


if __name__ == '__main__':
    main()
