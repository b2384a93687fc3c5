set -e
for n in 5120 7680 10240 11500 15360 20480; do echo "== single-chain helmet segments $n"; PROSPER_PT_DEBUG_SEGMENTS=$n python scripts/quick_bench.py --config helmet --single-chain --steps 8 | tail -2; done
for n in 1800 2560 3700 5120 7400; do echo "== pipelined helmet segments $n"; PROSPER_PT_DEBUG_SEGMENTS=$n python scripts/pipelined_bench.py helmet | tail -2; done
