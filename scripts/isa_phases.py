"""Static instruction mix between the s_memtime probes of a -DCN_PHASE_TIMING build (one kernel of a hipcc -S dump):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DCN_PHASE_TIMING -S --cuda-device-only \
          crowdnav_amd/csrc/crowdnav_amd.hip -o /tmp/ft.s && python scripts/isa_phases.py /tmp/ft.s rollout_fused_kernelILb1"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l.split(':')[0] and ':' in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = [l.strip().split(';')[0].rstrip() for l in lines[start + 1:end]
        if l.strip() and not l.strip().startswith(';') and not (l.strip().startswith('.') and not l.strip().split(';')[0].rstrip().endswith(':'))]
marks = [i for i, l in enumerate(body) if l.startswith('s_memtime')]
print(len(body), 'instructions; probes at', marks)
for a, b in zip(marks[:-1], marks[1:]):
    seg = body[a:b]
    c = lambda p: sum(1 for l in seg if re.match(p, l))  # noqa: E731
    print('%5d instr  valu %4d (trans %2d, f64 %3d)  salu %3d  lds %3d  vmem %2d  branch %2d  waitcnt %2d  readlane %2d  nop %2d' % (
        b - a, c(r'v_(?!readlane|writelane)'), c(r'v_(rcp|rsq|sqrt)'), c(r'v_\w+_f64'), c(r's_(?!cbranch|branch|waitcnt|nop|memtime)'),
        c(r'ds_'), c(r'(global|flat|scratch)_'), c(r's_c?branch'), c(r's_waitcnt'), c(r'v_(readlane|writelane)'), c(r's_nop')))
