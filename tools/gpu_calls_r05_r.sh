#!/bin/bash
# after the cancellation-free (1 - sigmoid) and the batched range tests: the failed fuzz seed, the GPU suite, the accuracy survey, config 3 timing
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 120 python tools/fuzz_parity.py --seeds 2000473 2000474 --verbose > $O/fuzz_2000473.txt 2>&1; grep -o '"passed": [0-9]*, "declined": [0-9]*, "failed": [0-9]*' $O/fuzz_2000473.txt | tail -1
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_suite_r.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite_r.log; tail -5 $O/gpu_suite_r.log | cut -c1-300
timeout -k 10 500 python tools/accuracy_survey.py 2048 $O/accuracy_survey_r > $O/accuracy_survey_r.txt 2> $O/accuracy_survey_r.err; tail -14 $O/accuracy_survey_r.txt | cut -c1-330
python bench.py --steps 1000 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 %8.2f us' % (j['ms_per_step']*1e3), 'auto %8.2f' % (j['solve_auto']['ms_per_step']*1e3), 'config2 %8.2f' % (j['secondary']['ms_per_step']*1e3))"
