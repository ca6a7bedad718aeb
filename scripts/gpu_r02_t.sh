# is the transitions atomic visible in short launches of the fused kernel?  launch probe: product build vs a build without it
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/r02t; mkdir -p $OUT; cd $REPO
echo "== product"; timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_product.txt
echo "== no transitions atomic"; CROWDNAV_AMD_LIB=$REPO/crowdnav_amd/lib/exp/lib_noatomic.so timeout 150 python scripts/launch_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_noatomic.txt
