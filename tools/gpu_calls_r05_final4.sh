#!/bin/bash
# Round-5 evidence at the hash of 5a65afb, call 4 (short): the fuzz slice of the GPU suite and a minute of the campaign with the
# harness's side-check tally (the harness changed after call 1 ran the suite), then the host-boundary figures at this hash:
# the PCIe-inclusive step (never `value`) and the class surface's one-robot latency.
O=gpurun_out/r05/final; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x > $O/gpu_fuzz_slice.log 2>&1; rc=$?; tail -3 $O/gpu_fuzz_slice.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python tools/fuzz_parity.py --seeds 9000000 9100000 --minutes 1 > $O/fuzz_parity_tally.json 2>&1; grep -A8 side_checks $O/fuzz_parity_tally.json | cut -c1-200; grep -E "\"cases\"|\"passed\"|\"declined\"|\"failed\"" $O/fuzz_parity_tally.json | tr -d '\n'; echo
(python tools/pcie_inclusive.py config2 4096; python tools/pcie_inclusive.py config3 65536) > $O/pcie_inclusive.txt 2> /dev/null; cat $O/pcie_inclusive.txt
(python tools/dropin_latency.py 7 300; python tools/dropin_latency.py 32 300) > $O/dropin_latency.txt 2> /dev/null; cat $O/dropin_latency.txt | cut -c1-300
