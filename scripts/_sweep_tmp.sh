for order in 1 0; do
  if [ $order = 0 ]; then export PROSPER_PT_DEBUG_NO_TILE_ORDER=1; echo "== raster order (PROSPER_PT_DEBUG_NO_TILE_ORDER=1)"; else echo "== tiles by cost"; fi
  for c in helmet c3 c4 c2; do python scripts/quick_bench.py --config $c --single-chain --steps 6 | tail -2 | head -1; done
  python scripts/pipelined_bench.py helmet c3 c4 c2 | grep "ranks 1 pipelined"
done
