# repeat the GPU parity suite to catch rare races (solver hand-offs, LDS-only barriers, DMA ring)
n=${N:-15}; fail=0
for i in $(seq 1 $n); do
  out=$(timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1)
  echo "run $i: $out"
  case "$out" in *failed*|*error*) fail=$((fail+1));; esac
done
echo "soak: $fail failing runs of $n"
