#!/bin/bash
# soak of the fuzz campaign at the round's final kernels, fresh seeds
O=gpurun_out/r05/final; mkdir -p $O
timeout -k 10 1080 python tools/fuzz_parity.py --seeds 5000000 5200000 --minutes 16.5 --log $O/fuzz_parity_soak.log > $O/fuzz_parity_soak.json 2>&1; tail -45 $O/fuzz_parity_soak.json | cut -c1-200
