"""bench.measure_sarl called several times in one process: argv = sequence of 0/1 (with_om)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for a in sys.argv[1:]:
    r = bench.measure_sarl(4096, 5, a == '1', 50, 10, 30, 1, 0, 0)
    print('om', a, 'select ms %.3f step ms %.3f' % (r['roofline']['select_ms'], r['roofline']['step_ms']), flush=True)
