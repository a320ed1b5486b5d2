#!/usr/bin/env python3
"""Static instruction mix of a kernel in a hipcc -S listing: tools/isa_count.py file.s kernel_substring [...]"""
import re
import sys
from collections import Counter


def count(path, names):
    s = open(path).read()
    out = []
    for name in names:
        m = re.search(r'^(_Z\S*' + re.escape(name) + r'\S*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M)
        if not m:
            out.append((name, None))
            continue
        ins = [l.split()[0] for l in m.group(2).splitlines() if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        out.append((name, dict(total=len(ins), valu=valu, mfma=sum(v for k, v in c.items() if k.startswith('v_mfma')),
                               lds=sum(v for k, v in c.items() if k.startswith('ds_')),
                               vmem=sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))),
                               salu=sum(v for k, v in c.items() if k.startswith('s_')), top=c.most_common(12))))
    return out


if __name__ == "__main__":
    for name, r in count(sys.argv[1], sys.argv[2:]):
        print(name, r)
