#!/bin/bash
set -e
mkdir -p gpurun_out
OUT=gpurun_out/closest_variants.txt
: > $OUT
python - >> $OUT 2>&1 <<'PY'
import torch
x = torch.empty(403 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("zero_ (memset) 403 MiB", lambda: x.zero_()), ("copy_ 403 MiB read + 403 MiB written", lambda: y.copy_(x))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: {us:.1f} us = {x.numel() * 4 / us / 1e6:.2f} TB/s written")
PY
for v in main cr8 cr32 cr64 nt nt32; do
  echo "== $v" >> $OUT
  if [ $v = main ]; then python tools/closest_stage_timing.py 65536 50 >> $OUT 2>&1
  else RMP2_LIB=tools/diag/librmp2_$v.so python tools/closest_stage_timing.py 65536 50 >> $OUT 2>&1; fi
done
grep -v amdgpu.ids $OUT
