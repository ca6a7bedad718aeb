"""MFMA-pipe occupancy of the value-network kernels from one rocprofv3 --pmc pass each (scripts/gpu.sh pmcnet):
    busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMD-cycles of the kernel) = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
(GRBM_GUI_ACTIVE is summed over the 8 XCDs; round 5's check: 3.697 G / (33.36 M / 8 x 1024) = 0.866 for sarl_reg_kernel) — the
share of the kernel's time the matrix pipe is executing an MFMA, padding k-steps and padded tiles included.  Beside it the kernel's
average duration from the kernel trace of the same command and the executed FP32-MFMA rate (SQ_INSTS_MFMA x 512 flop x 4 ... per
v_mfma_f32_16x16x4_f32: 16 x 16 x 4 x 2 = 2048 flop).

    python scripts/mfma_busy.py gpurun_out/r06
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for v in ('sarl', 'om_sarl', 'cadrl', 'lstm_rl', 'lstm_rl2'):
    pm = glob.glob(os.path.join(root, 'pmc_%s_mfma' % v, '**', '*counter_collection.csv'), recursive=True)
    tr = glob.glob(os.path.join(root, 'trace_%s' % v, '**', '*kernel_stats.csv'), recursive=True)
    if not pm:
        continue
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(pm[0])):
        k = row['Kernel_Name'].split('(')[0]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        n[k][row['Counter_Name']] += 1
    dur = {}
    if tr:
        for row in csv.DictReader(open(tr[0])):
            dur[row['Name'].split('(')[0]] = float(row['AverageNs'])
    for k in acc:
        c = {name: val / n[k][name] for name, val in acc[k].items()}
        if c.get('SQ_INSTS_MFMA', 0) <= 0:
            continue
        simd_cycles = c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
        busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
        line = '%-9s %-48s MFMA-pipe busy %.3f  (SQ_VALU_MFMA_BUSY_CYCLES %.4g / SIMD-cycles %.4g)  MFMAs %.4g' % (
            v, k[:48], busy, c['SQ_VALU_MFMA_BUSY_CYCLES'], simd_cycles, c['SQ_INSTS_MFMA'])
        if k in dur:
            line += '  avg %.3f ms  executed %.1f TFLOP/s (%.3f of 157.3)' % (dur[k] / 1e6, c['SQ_INSTS_MFMA'] * 2048 / dur[k] / 1e3,
                                                                          c['SQ_INSTS_MFMA'] * 2048 / dur[k] / 1e3 / 157.3)
        print(line)
