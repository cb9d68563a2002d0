#!/bin/bash
# end-of-round checks on one GPU box: the -m gpu suite, the quad-forced suites under every register cap and in the general
# form, smoke(), and the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in 2 3 4; do
RMP2_KERNEL=quad RMP2_QUAD_MINW=$w timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py tests/test_gpu_random_robots.py -m gpu -x -q 2>&1 | tail -1
done
RMP2_KERNEL=quad RMP2_QUAD_SYM=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -1
RMP2_KERNEL=hex timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py -m gpu -x -q 2>&1 | tail -1
RMP2_KERNEL=lane timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capsules.py -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-200
