"""Seconds per part of the single-episode RL sampling calls (Explorer.rl_profile): weights + reset / the step stream / read-back."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import crowdnav_amd.compat.explorer as ex
orig = ex.Explorer.__init__
prof = {}
def init(self, *a, **k):
    orig(self, *a, **k)
    if os.environ.get('CN_RL_PARTS_NOPROF') != '1':  # (1: no laps, no extra synchronisation — only the call's own time below)
        self.rl_profile = prof
ex.Explorer.__init__ = init
spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
args = mod.parser().parse_args(['--gpu', '--il-episodes', '100', '--il-epochs', '2', '--train-episodes', os.environ.get('CN_RL_PARTS_EPISODES', '300'), '--train-batches', os.environ.get('CN_RL_PARTS_BATCHES', '1'),
                                '--evaluation-interval', '100000', '--val-size', '4', '--test-size', '4', '--seed', '0'])
if os.environ.get('CN_RL_PARTS_GC') == 'off':  # (is the cyclic collector what the parts pay "in situ"?)
    import gc
    gc.disable()
elif os.environ.get('CN_RL_PARTS_GC') == 'freeze':
    import gc
    gc.collect(); gc.freeze()
out = mod.run(args)
n = int(os.environ.get('CN_RL_PARTS_EPISODES', '300'))
print({k: round(v / n * 1e3, 3) if isinstance(v, float) else v for k, v in prof.items()}, '(ms per call)', 'env steps', out['timing']['rl_env_steps'], 'rl_sample_s per episode: %.3f ms' % (out['timing']['rl_sample_s'] / n * 1e3))
