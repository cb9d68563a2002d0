#!/bin/bash
mkdir -p gpurun_out
RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=quad python tools/stamps.py 65536 > gpurun_out/stamps_link.txt 2>&1
grep -v amdgpu.ids gpurun_out/stamps_link.txt | tail -12
