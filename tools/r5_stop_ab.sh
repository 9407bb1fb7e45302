cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in default $PWD/tools/ubench/libzoic_plainstop.so; do
  if [ "$lib" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$lib; fi
  for c in C2 C3 C4 C5; do
    for p in fast unchecked; do
    python bench.py --only-headline --config $c --precision $p --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rep $lib $c $p', d['value'], d['ms_per_step'])"
    done
  done
done
done
