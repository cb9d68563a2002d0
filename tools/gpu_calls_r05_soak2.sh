#!/bin/bash
# soak of the fuzz campaign at the round's LAST kernel hash (after the final dispatch commit), fresh seeds, then the pair grid
O=gpurun_out/r05/final; mkdir -p $O
M=${FUZZ_MINUTES:-14}
timeout -k 10 $((M * 60 + 90)) python tools/fuzz_parity.py --seeds 7000000 7200000 --minutes $M --log $O/fuzz_parity_soak.log > $O/fuzz_parity_soak.json 2>&1; tail -45 $O/fuzz_parity_soak.json | cut -c1-200
timeout -k 10 100 python tools/fuzz_parity.py --pairs --seeds 4000 4400 --minutes 1.2 > $O/fuzz_parity_pair_grid.json 2>&1; tail -12 $O/fuzz_parity_pair_grid.json | cut -c1-200
