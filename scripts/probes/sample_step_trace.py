"""One policy's streamed sampling steps (bench.measure_sample_step) for a rocprofv3 --kernel-trace --stats run:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o trace -- python scripts/probes/sample_step_trace.py [sarl|lstm_rl] [om]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

policy = sys.argv[1] if len(sys.argv) > 1 else 'sarl'
om = len(sys.argv) > 2
r = bench.measure_sample_step(0, with_om=om, policy=policy)
print('%s om=%d %.1f us per step' % (policy, om, r['us_per_step']))
