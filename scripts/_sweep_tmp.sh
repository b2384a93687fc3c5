for lib in build/variants/lib_tri1.so build/variants/lib_tri2.so build/variants/lib_refill24.so build/variants/lib_cont2.so; do
  echo "== lib $lib"
  for c in c4 c3 c2; do
    python scripts/quick_bench.py --config $c --single-chain --steps 6 --lib $lib | tail -2 | head -1
  done
done
