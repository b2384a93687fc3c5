for c in c2 c3 c4 helmet; do python scripts/quick_bench.py --config $c --single-chain --steps 6 | tail -2; done
python scripts/pipelined_bench.py helmet c3 c4 | grep pipelined
