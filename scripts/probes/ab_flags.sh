cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" build/exp/lib_ab_*.so; do
  n=$(basename "${lib:-intree}" .so)
  echo -n "$n default: "; CROWDNAV_AMD_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --steps 4000 2>/dev/null | python scripts/bench_line.py /dev/stdin
  echo -n "$n h20: "; CROWDNAV_AMD_LIB=$lib python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 2997 --warmup 999 --chunk 999 2>/dev/null | python scripts/bench_line.py /dev/stdin
done; done
