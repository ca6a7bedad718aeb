# boundary_ms of the driver's shape, in-tree library against build/exp/lib_ab_*.so (several runs each)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for lib in "" build/exp/lib_ab_*.so; do
  echo -n "${lib:-intree}: "; CROWDNAV_AMD_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --no-r3-definition --no-fill-probe --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value %.1f M  boundary %.1f us  incl %.1f M  launch %.1f us' % (d['value'] / 1e6, d['boundary_ms'] * 1e3, d['value_incl_boundary'] / 1e6, d['roofline']['avg_launch_ms'] * 1e3))"
done; done
