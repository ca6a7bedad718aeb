"""cProfile of the single-episode RL sampling calls of examples/train_sarl.py (host side: where an episode's 5 ms go)."""
import cProfile, importlib.util, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
args = mod.parser().parse_args(['--gpu', '--il-episodes', '100', '--il-epochs', '2', '--train-episodes', '300', '--train-batches', '1',
                                '--evaluation-interval', '100000', '--val-size', '4', '--test-size', '4', '--seed', '0'])
pr = cProfile.Profile(); pr.enable(); out = mod.run(args); pr.disable()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out['timing'].items()})
st = pstats.Stats(pr); st.sort_stats('cumulative')
st.print_stats('explorer|engine|_lib|memory|trainer', 28)
