#!/bin/bash
# round 3, GPU call 3: which shards carry robots on the careful resolve path, and what that costs the launch
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python tools/flag_tail.py > $O/flag_tail.txt 2> $O/flag_tail.err; cat $O/flag_tail.txt; tail -3 $O/flag_tail.err
timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline > $O/bench_config3b.json 2> $O/bench_config3b.err; cut -c1-1500 $O/bench_config3b.json; tail -2 $O/bench_config3b.err
