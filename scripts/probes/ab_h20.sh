cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in "" build/exp/lib_ab_*.so; do
  echo -n "${lib:-intree}: "; CROWDNAV_AMD_LIB=$lib python bench.py --no-cpu-baseline --humans 20 --circle-radius 12 --steps 3996 --warmup 999 --chunk 999 2>/dev/null | python scripts/bench_line.py /dev/stdin
done; done
