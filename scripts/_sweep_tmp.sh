for raw in 0 1; do echo "== PROSPER_PT_DEBUG_RAW_RECORDS=$raw"; for c in c3 c4 c2 helmet; do PROSPER_PT_DEBUG_RAW_RECORDS=$raw python scripts/quick_bench.py --config $c --single-chain --steps 6 | tail -2 | head -1; done; done
for raw in 0 1; do echo "== pipelined, RAW=$raw"; PROSPER_PT_DEBUG_RAW_RECORDS=$raw python scripts/pipelined_bench.py c3 c4 | grep pipelined; done
