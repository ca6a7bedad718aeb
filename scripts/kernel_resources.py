"""Per-kernel register / spill / LDS table of libcrowdnav_amd.so's code object, from hipcc's own resource report
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python scripts/kernel_resources.py [filter-regex] [-D...]      # extra -D flags go to hipcc
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'crowdnav_amd', 'csrc', 'crowdnav_amd.hip')


def report(extra=()):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-slp-vectorize',
           '-fno-fast-math', '-Rpass-analysis=kernel-resource-usage', '-c', SRC, '-o', '/dev/null'] + list(extra)
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, cwd=os.path.dirname(SRC)).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: +(.*?): (.*?) \[-Rpass', line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ('Function Name', 'Name'):
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def demangle(names):
    out = subprocess.run(['c++filt'] + names, stdout=subprocess.PIPE, text=True).stdout
    return [re.sub(r'\(.*', '', n).replace('void ', '') for n in out.splitlines()]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('-D')]
    extra = [a for a in sys.argv[1:] if a.startswith('-D')]
    pat = re.compile(args[0]) if args else None
    rows = report(extra)
    if not rows:
        sys.exit('no kernel resource remarks: the compile failed (run the hipcc command by hand to see why)')
    names = demangle([r['name'] for r in rows])
    print('%-52s %5s %5s %5s %7s %7s %8s %4s %7s' % ('kernel', 'SGPR', 'VGPR', 'AGPR', 'sp.SGPR', 'sp.VGPR', 'scratch', 'occ', 'LDS'))
    for r, n in zip(rows, names):
        if pat and not pat.search(n):
            continue
        g = lambda *ks: next((r[k] for k in ks if k in r), '-')  # noqa: E731
        print('%-52s %5s %5s %5s %7s %7s %8s %4s %7s' % (
            n[-52:], g('TotalSGPRs', 'SGPRs'), g('VGPRs'), g('AGPRs'), g('SGPRs Spill'), g('VGPRs Spill'),
            g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')))


if __name__ == '__main__':
    main()
