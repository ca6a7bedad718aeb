"""cProfile of the single-episode RL sampling calls (Explorer.run_k_episodes(1, 'train', update_memory=True)) inside
examples/train_sarl.py --gpu: where the host time of a sampled episode goes OUTSIDE the device work."""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import crowdnav_amd.compat.explorer as ex  # noqa: E402

prof = cProfile.Profile()
orig = ex.Explorer.run_k_episodes


def wrapped(self, k, phase, update_memory=False, **kw):
    if phase == 'train' and update_memory and not kw.get('imitation_learning'):
        prof.enable()
        try:
            return orig(self, k, phase, update_memory=update_memory, **kw)
        finally:
            prof.disable()
    return orig(self, k, phase, update_memory=update_memory, **kw)


ex.Explorer.run_k_episodes = wrapped
spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
n = 300
args = mod.parser().parse_args(['--gpu', '--il-episodes', '100', '--il-epochs', '2', '--train-episodes', str(n), '--train-batches', '1',
                                '--evaluation-interval', '100000', '--val-size', '4', '--test-size', '4', '--seed', '0'])
out = mod.run(args)
print('rl_sample_s per episode: %.3f ms' % (out['timing']['rl_sample_s'] / n * 1e3))
st = pstats.Stats(prof)
st.sort_stats('tottime').print_stats(22)
