# timing-only ablations of the value-network kernel (results are wrong on purpose): what does a layer's fixed cost consist of?
cd /tmp && export TMPDIR=/tmp; REPO=$GRAFT_REPO_ROOT; cd $REPO
for v in "" NO_B NO_A NO_EPILOGUE NO_BARRIER MFMA_ONLY; do
  for pipe in 1 0; do
    if [ -z "$v" ]; then L=""; else L=$REPO/crowdnav_amd/lib/exp/lib_sarl_$v.so; fi
    echo -n "variant ${v:-product} pipe $pipe: "; CROWDNAV_AMD_LIB=$L CROWDNAV_AMD_SARL_PIPE=$pipe timeout 120 python scripts/sarl_bench.py 2>&1 | grep with_om
  done
done
