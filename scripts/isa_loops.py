"""Static instruction mix of the loops of one kernel in a `hipcc -S --cuda-device-only` dump: for every backward branch
(label .. branch), the instruction classes inside — in particular v_readlane / v_writelane (SGPR spills living in VGPR
lanes) inside the step loop, which the whole-kernel spill count of the resource report cannot tell from prologue spills.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S --cuda-device-only \
          crowdnav_amd/csrc/crowdnav_amd.hip -o /tmp/cn.s && python scripts/isa_loops.py /tmp/cn.s rollout_fused_kernelILb1 [min_size]
"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0] and ':' in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body, labels = [], {}
for l in lines[start + 1:end + 1]:
    t = l.strip().split(';')[0].rstrip()
    if not t:
        continue
    if t.endswith(':'):
        labels[t[:-1]] = len(body)
        continue
    if t.startswith('.'):
        continue
    body.append(t)


def mix(seg):
    c = lambda p: sum(1 for l in seg if re.match(p, l))  # noqa: E731
    return ('%5d instr  valu %4d (trans %2d, f64 %3d)  salu %3d  lds %3d  vmem %2d  smem %2d  branch %3d  waitcnt %3d  '
            'readlane %3d writelane %3d  nop %2d' % (
                len(seg), c(r'v_(?!readlane|writelane)'), c(r'v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup)'), c(r'v_\w+_f64'),
                c(r's_(?!cbranch|branch|waitcnt|nop|load|buffer_load)'), c(r'ds_'), c(r'(global|flat|scratch|buffer)_'),
                c(r's_(load|buffer_load)'), c(r's_c?branch'), c(r's_waitcnt'), c(r'v_readlane'), c(r'v_writelane'), c(r's_nop')))


print('%d instructions in %s' % (len(body), key))
print('whole kernel:', mix(body))
loops = []
for i, t in enumerate(body):
    m = re.match(r's_c?branch\w*\s+(\S+)', t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
for a, b, name in sorted(loops, key=lambda x: x[0] - x[1]):
    if b - a >= min_size:
        print('loop %-12s [%6d..%6d]:' % (name, a, b), mix(body[a:b + 1]))
