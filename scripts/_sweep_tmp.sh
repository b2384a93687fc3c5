for rep in 1 2; do for s in 1216 1344 1472 1600 1728 1856 1984 2112; do echo "rep $rep seglen $s"; PROSPER_PT_DEBUG_SEGLEN=$s python scripts/quick_bench.py --steps 20 | tail -1 || exit 1; done; done
