#!/bin/bash
mkdir -p gpurun_out
python tools/diag/link_debug.py > gpurun_out/link_debug.txt 2>&1
grep -v amdgpu.ids gpurun_out/link_debug.txt | tail -40
