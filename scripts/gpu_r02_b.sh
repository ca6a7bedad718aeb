# launch-cost probe: in-tree build vs experimental builds
mkdir -p gpurun_out/r02b && cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02b; cd $REPO
timeout 300 python -m pytest tests/test_boundary.py -m gpu -q -x 2>&1 | tail -n 3
echo "== base"; timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_base.txt
for v in "$@"; do echo "== $v"; CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/$v timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_$v.txt; done
echo "== base E=1"; CROWDNAV_AMD_ENVS_PER_WAVE=1 timeout 200 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_e1.txt
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -n 1 | cut -c 1-200; done
