"""secondary.sample_step's rows on their own (bench.measure_sample_step): SARL / OM-SARL / LSTM-RL / LSTM-RL with maps, one env,
and the same with CROWDNAV_AMD_SARL_NARROW=0 (the general route inside the call) for comparison."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

for narrow in ('1', '0'):
    os.environ['CROWDNAV_AMD_SARL_NARROW'] = narrow
    for policy, om in (('sarl', False), ('sarl', True), ('lstm_rl', False), ('lstm_rl', True)):
        r = bench.measure_sample_step(0, with_om=om, policy=policy)
        print('NARROW=%s %-8s om=%d  %.1f us per step, launches %s %s' % (narrow, policy, om, r['us_per_step'], r['launches_per_step'], r['launch_counts']))
