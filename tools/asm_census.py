#!/usr/bin/env python3
"""Instruction census of one kernel in a device assembly listing (hipcc -S --cuda-device-only): counts by mnemonic, float64 share.
  hipcc --offload-arch=gfx950 <flags> -S --cuda-device-only -o /tmp/decode.s csrc/decode.hip; python tools/asm_census.py /tmp/decode.s <mangled-name-substring>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m.group(1) for m in re.finditer(r'^(\S+):\s*; @', txt, re.M) if pat in m.group(1)]
for name in names:
    i = txt.index(name + ':')
    j = txt.index('.Lfunc_end', i)
    ops = collections.Counter()
    for ln in txt[i:j].splitlines():
        m = re.match(r'\s+([vs]_[a-z0-9_]+|buffer_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|flat_[a-z0-9_]+)', ln)
        if m:
            ops[m.group(1)] += 1
    f64 = {k: v for k, v in ops.items() if 'f64' in k}
    print(name)
    print("  total", sum(ops.values()), "| VALU", sum(v for k, v in ops.items() if k.startswith('v_')), "| SALU", sum(v for k, v in ops.items() if k.startswith('s_')),
          "| float64", sum(f64.values()), "| vmem", sum(v for k, v in ops.items() if k.startswith(('buffer_', 'global_', 'flat_'))), "| lds", sum(v for k, v in ops.items() if k.startswith('ds_')))
    print("  f64:", dict(sorted(f64.items(), key=lambda kv: -kv[1])))
    print("  top:", ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
