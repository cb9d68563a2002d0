#!/bin/bash
# Round-5 evidence at the hash of 5a65afb, call 5: what is left of the round's GPU budget as a second soak on fresh seeds
O=gpurun_out/r05/final; mkdir -p $O
M=${FUZZ_MINUTES:-6}
timeout -k 10 $((M * 60 + 60)) python tools/fuzz_parity.py --seeds 10000000 10200000 --minutes $M --log $O/fuzz_parity_soak_b.log > $O/fuzz_parity_soak_b.json 2>&1; tail -50 $O/fuzz_parity_soak_b.json | cut -c1-200
